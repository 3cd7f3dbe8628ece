# rocprofv3 kernel trace of the default bench workload: prof_bench.sh [tag]  ->  gpurun_out/prof_<tag>/ + gpurun_out/prof_<tag>_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05_final}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-contract-ab --no-stress > $R/gpurun_out/prof_${T}_bench.log 2>&1
# the bench's JSON line, NOT the log's last line (rocprofv3 writes its own finalisation message behind it)
grep '^{"metric"' $R/gpurun_out/prof_${T}_bench.log | tail -1 > $R/gpurun_out/prof_${T}_bench.json
cut -c1-200 $R/gpurun_out/prof_${T}_bench.json
python $R/scripts/split_kernel_stats.py "$(find $R/gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/prof_${T}_kernel_stats.csv   # counting variants below a separator
head -24 $R/gpurun_out/prof_${T}_kernel_stats.csv | cut -c1-160
