cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### pytest -m gpu"; python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_call6.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03_pytest_call6.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/r03_pytest_call6.log | head -20
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],1), 'odo', d['stage_ms']['odometry'], 'pipe', d.get('stage_ms_pipelined'), d.get('planned_frames'), 'frac', round(r['frac'],4), 'alone', round(r['frac_alone'],4), 'in-frame us', round(1e3*r['avg_launch_ms'],1), 'alone us', round(1e3*r['avg_launch_ms_alone'],1), 'lane_eff', r.get('lane_efficiency'), d['config']['frame_ms']['p50'])"; }
run() { KT_HIP_LIB=$2 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "$1"; }
for rep in 1 2; do
  run "tree" ""
  run "exact_end=0" $PWD/kintinuous_amd/libkt_exp_1.so
  run "sweeper=0" $PWD/kintinuous_amd/libkt_exp_2.so
  run "rec16" $PWD/kintinuous_amd/libkt_exp_3.so
done
echo "#### farwall"; for l in "" $PWD/kintinuous_amd/libkt_exp_1.so; do KT_HIP_LIB=$l python bench.py --workload farwall768 --steps 20 --warmup 4 --no-cpu-baseline --no-stress --no-readahead 2>/dev/null | line "farwall[$l]"; done
echo "#### crabwalk"; for l in "" $PWD/kintinuous_amd/libkt_exp_2.so; do KT_HIP_LIB=$l python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab[$l]"; done
echo "#### slice stage kernels"; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_slice -- python $GRAFT_REPO_ROOT/scripts/slice_stage_timing.py > $GRAFT_REPO_ROOT/gpurun_out/r03_slice_stage_timing.md 2>&1); tail -4 gpurun_out/r03_slice_stage_timing.md; f=$(ls -t gpurun_out/prof_slice/*/*kernel_stats.csv | head -1); grep -i "slice\|rocprim\|radix\|scan" $f | cut -c1-150 | head -12
