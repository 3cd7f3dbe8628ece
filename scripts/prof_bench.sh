cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_v12 -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_v12_bench.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_r01_v12_bench.log | cut -c1-200
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01_v12 -name "*kernel_stats.csv" | head -2
