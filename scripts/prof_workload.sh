# rocprofv3 kernel trace of one workload: prof_workload.sh <workload> <steps> <tag>  ->  gpurun_out/prof_<tag>_kernel_stats.csv + prof_<tag>_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$3 -- python $R/bench.py --workload $1 --steps $2 --warmup 10 --no-cpu-baseline --no-contract-ab > $R/gpurun_out/prof_$3.log 2>&1
grep '^{"metric"' $R/gpurun_out/prof_$3.log | tail -1 > $R/gpurun_out/prof_$3_bench.json
cut -c1-150 $R/gpurun_out/prof_$3_bench.json
python $R/scripts/split_kernel_stats.py "$(find $R/gpurun_out/prof_$3 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/prof_$3_kernel_stats.csv
cut -c1-150 $R/gpurun_out/prof_$3_kernel_stats.csv | head -24
