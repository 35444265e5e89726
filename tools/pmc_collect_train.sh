#!/bin/bash
# HBM traffic of the training-step kernels (run on the GPU box through gpurun):
#   tools/pmc_collect_train.sh <tag>  ->  gpurun_out/<tag>/{fetch,write,tcc}_counter_collection.csv ; summarise with
#   python tools/pmc_summary.py --train gpurun_out/<tag> <prefix>
set -u
TAG=${1:-pmc_train}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python "$R/tools/train_kernels_probe.py" 2 > "$OUT/log_$name.txt" 2>&1
  echo "$name exit $?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
rm -f "$OUT"/*_kernel_trace.csv
