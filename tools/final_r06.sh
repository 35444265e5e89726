# Round-6 evidence run (through gpurun): rocprofv3 kernel statistics of the headline bench command, of the cfg 5 training step and
# of the cfg 3 loop, the timeline of one training step, then the PMC passes (separate runs, --kernel-trace only).  Text tables go
# to gpurun_out/r06/ and are copied into profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b /tmp/prof_t /tmp/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-cfg3 --no-cfg5 --no-alt --no-variants --no-live-pmc --repeats 5 > $O/stats_bench_line.txt 2>&1
f=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_summary.py $f $O/r06_kernel_stats.txt adam_step > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o train -- python $R/tools/train_probe.py --views-in 32 --views-out 8 --amp --steps 3 > $O/stats_train_line.txt 2>&1
f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1)
python $R/tools/kernel_stats_txt.py $f $O/r06_train_step_kernel_stats.txt "cfg 5 training step: 32 + 8 views, SYN(128,16), bf16 autocast + bf16 storage (tools/train_probe.py --views-in 32 --views-out 8 --amp --steps 3: 4 steps incl. warm-up)" 4 > /dev/null
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f $O/r06_train_step_timeline_full.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o cfg3 -- python $R/tools/cfg3_probe.py 10 > $O/stats_cfg3_line.txt 2>&1
f=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1)
python $R/tools/kernel_stats_txt.py $f $O/r06_released_arch_kernel_stats.txt "BASELINE cfg 3 (released architecture, cross_entropy_linemod, 128 renders per iteration): tools/cfg3_probe.py 10 = 30 iterations + 2 reconstructions" 30 > /dev/null
cd $R
bash tools/occ_profile.sh r06 > $O/log_occ.txt 2>&1
cd $R
bash tools/pmc_collect.sh r06_pmc > $O/log_pmc1.txt 2>&1
bash tools/pmc_collect_train.sh r06_pmc_train > $O/log_pmc2.txt 2>&1
bash tools/pmc_collect_cfg3.sh r06_pmc_cfg3 > $O/log_pmc3.txt 2>&1
find gpurun_out/r06_pmc gpurun_out/r06_pmc_train gpurun_out/r06_pmc_cfg3 -name "*_kernel_trace.csv" ! -name "trace_*" -delete
python tools/pmc_summary.py gpurun_out/r06_pmc $O/r06 > /dev/null 2>&1
python tools/pmc_summary.py --train gpurun_out/r06_pmc_train $O/r06_train > /dev/null 2>&1
python tools/pmc_summary.py --cfg3 gpurun_out/r06_pmc_cfg3 $O/r06_released_arch > /dev/null 2>&1
rm -rf gpurun_out/r06_pmc gpurun_out/r06_pmc_train gpurun_out/r06_pmc_cfg3
ls -la $O | tail -20
tail -2 $O/stats_train_line.txt | cut -c1-300
