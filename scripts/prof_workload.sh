# rocprofv3 kernel trace of one workload: prof_workload.sh <workload> <steps> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$3 -- python $R/bench.py --workload $1 --steps $2 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_$3.log 2>&1
tail -1 $R/gpurun_out/prof_$3.log | cut -c1-150
cat $(find $R/gpurun_out/prof_$3 -name "*kernel_stats.csv" | head -1) | cut -c1-150 | head -24
