#!/bin/bash
# rocprofv3 evidence for BASELINE cfg 3 (released architecture, cross_entropy_linemod on the fused engine):
#   tools/pmc_collect_cfg3.sh <tag>   ->  gpurun_out/<tag>/{trace,sq1,sq2,fetch,write}_*.csv
# kernel trace first, then the counter passes on their own (--kernel-trace + --pmc only).
set -u
TAG=${1:-cfg3_pmc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python "$R/tools/cfg3_probe.py" 10 > "$OUT/log_trace.txt" 2>&1
echo "trace exit $?"
run() {
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python "$R/tools/cfg3_probe.py" 3 > "$OUT/log_$name.txt" 2>&1
  echo "$name exit $?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
ls "$OUT" | head -30
