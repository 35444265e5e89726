# rocprofv3 kernel statistics of the occlusion renderer variant's pose iteration (tools/variant_probe.py --only occlusion): text table
# into gpurun_out/r06/<tag>_occlusion_variant_kernel_stats.txt.   bash tools/occ_profile.sh [tag=r06]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_o
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o -o occ -- python $R/tools/variant_probe.py --only occlusion --engine-only --iters 10 > $O/occ_probe_line.txt 2>&1
f=$(find /tmp/prof_o -name "*kernel_stats.csv" | head -1)
python $R/tools/kernel_stats_txt.py $f $O/${TAG}_occlusion_variant_kernel_stats.txt "occlusion variant, explicit kernels: 13 evaluations SYN(128,16) N=8 (tools/variant_probe.py --only occlusion --engine-only --iters 10)" 13 > /dev/null
head -22 $O/${TAG}_occlusion_variant_kernel_stats.txt | cut -c1-150
