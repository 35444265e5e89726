# rocprofv3 kernel statistics of the headline bench command and of the cfg 5 training step (run through gpurun); the text tables go
# to gpurun_out/ and are copied into profiles/ by hand
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b /tmp/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-cfg3 --no-cfg5 --no-alt --no-variants --repeats 5 > $R/gpurun_out/stats_bench_line.txt 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
python $R/tools/kernel_stats_txt.py $f $R/gpurun_out/r05_kernel_stats.txt "bench.py --no-cpu-baseline --no-cfg3 --no-cfg5 --no-alt --no-variants --repeats 5 (headline loop: 3 warm-up + 5 x 20 timed iterations, the 16-view reconstruction twice, the 100-iteration reference-trace comparison)" 1 > /dev/null
cp $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r05_bench_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o train -- python $R/tools/train_probe.py --views-in 32 --views-out 8 --amp --steps 3 > $R/gpurun_out/stats_train_line.txt 2>&1
f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1)
python $R/tools/kernel_stats_txt.py $f $R/gpurun_out/r05_train_step_kernel_stats.txt "cfg 5 training step: 32 + 8 views, SYN(128,16), bf16 autocast + bf16 storage (tools/train_probe.py --views-in 32 --views-out 8 --amp --steps 3: 4 steps incl. warm-up)" 4 | head -30 | cut -c1-170
tail -2 $R/gpurun_out/stats_train_line.txt | cut -c1-400
