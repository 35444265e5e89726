set -u
cd $GRAFT_REPO_ROOT
bash tools/pmc_collect.sh r05_pmc > gpurun_out/log_pmc1.txt 2>&1
bash tools/pmc_collect_train.sh r05_pmc_train > gpurun_out/log_pmc2.txt 2>&1
bash tools/pmc_collect_cfg3.sh r05_pmc_cfg3 > gpurun_out/log_pmc3.txt 2>&1
# keep only the counter csv files (the kernel traces are large)
find gpurun_out/r05_pmc gpurun_out/r05_pmc_train gpurun_out/r05_pmc_cfg3 -name "*_kernel_trace.csv" ! -name "trace_*" -delete
du -sh gpurun_out/r05_pmc gpurun_out/r05_pmc_train gpurun_out/r05_pmc_cfg3
tail -3 gpurun_out/log_pmc1.txt gpurun_out/log_pmc2.txt gpurun_out/log_pmc3.txt
