#!/bin/bash
# PMC evidence for the heavy kernels of one pose iteration (run on the GPU box through gpurun):
#   tools/pmc_collect.sh <tag>      ->  gpurun_out/<tag>/{sq1,sq2,sq3,fetch,write,tcc}_counter_collection.csv
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots:
# 8 SQ counters per pass; FETCH_SIZE and WRITE_SIZE cannot share one).  Summarise with tools/pmc_summary.py.
set -u
TAG=${1:-pmc_r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python "$R/tools/hbm_probe.py" 2 > "$OUT/log_$name.txt" 2>&1
  echo "$name exit $?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VALU SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
ls -la "$OUT" | head -40
