cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### pytest subset"; python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_gpu_configs.py tests/test_gpu_tracker.py tests/test_slice_process.py tests/test_pcd.py tests/test_gpu_host_shell.py -m gpu -q > gpurun_out/r03_pytest_call7.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03_pytest_call7.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/r03_pytest_call7.log | head -20
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],1), 'odo', d['stage_ms']['odometry'], 'pipe', d.get('stage_ms_pipelined'), d.get('planned_frames'), 'frac', round(r['frac'],4), 'alone', round(r['frac_alone'],4), 'in-frame us', round(1e3*r['avg_launch_ms'],1), 'alone us', round(1e3*r['avg_launch_ms_alone'],1), 'lane_eff', r.get('lane_efficiency'), d['config']['frame_ms']['p50'])"; }
run() { KT_HIP_LIB=$2 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "$1"; }
for rep in 1 2; do
  run "tree" ""
  run "exact_end=0" $PWD/kintinuous_amd/libkt_exp_1.so
done
echo "#### farwall"; KT_HIP_LIB= python bench.py --workload farwall768 --steps 20 --warmup 4 --no-cpu-baseline --no-stress --no-readahead 2>/dev/null | line "farwall"
echo "#### kernel stats"; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03c -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress > $GRAFT_REPO_ROOT/gpurun_out/prof_r03c.log 2>&1); f=$(ls -t gpurun_out/prof_r03c/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r03_kernel_stats_plan3.csv; head -14 $f | cut -c1-140
python scripts/frame_timeline.py $(ls -t gpurun_out/prof_r03c/*/*kernel_trace.csv | head -1) | head -8
echo "#### slice stage kernels"; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_slice2 -- python $GRAFT_REPO_ROOT/scripts/slice_stage_timing.py > $GRAFT_REPO_ROOT/gpurun_out/r03_slice_stage_timing.md 2>&1); grep "^|" gpurun_out/r03_slice_stage_timing.md; f=$(ls -t gpurun_out/prof_slice2/*/*kernel_stats.csv | head -1); grep -i "slice\|rocprim" $f | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,60), $2}' | head -12
