# round 5, call 18: kernel tables of the other two workloads on the final tree
cd $GRAFT_REPO_ROOT
bash scripts/prof_workload.sh farwall768 40 r05_farwall768 > /dev/null 2>&1; head -7 gpurun_out/prof_r05_farwall768_kernel_stats.csv | cut -c1-150; cut -c1-120 gpurun_out/prof_r05_farwall768_bench.json
bash scripts/prof_workload.sh crabwalk512 120 r05_crabwalk512 > /dev/null 2>&1; head -5 gpurun_out/prof_r05_crabwalk512_kernel_stats.csv | cut -c1-150
