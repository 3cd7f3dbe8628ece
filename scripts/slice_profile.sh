# rocprofv3 per-kernel table of the per-slice stage (f2) on a real extracted slab -> gpurun_out/r04_slice_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_slice -- python $R/scripts/slice_stage_timing.py > $R/gpurun_out/r04_slice_stage_timing.md 2> $R/gpurun_out/prof_slice.err
cp "$(find $R/gpurun_out/prof_slice -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r04_slice_kernel_stats.csv
head -30 $R/gpurun_out/r04_slice_kernel_stats.csv | cut -c1-170
tail -12 $R/gpurun_out/r04_slice_stage_timing.md
