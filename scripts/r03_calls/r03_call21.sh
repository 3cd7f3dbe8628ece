cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### volume / tracker / configs tests (plan enqueued before the odometry)"
python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_gpu_tracker.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/call21_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call21_tests.log | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'serial odo', d['stage_ms']['odometry'], 'integ', d['stage_ms']['integrate'], 'tsdf23 %.1f us alone %.1f frac %.4f stress %.4f ms' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], s.get('avg_launch_ms', 0)), 'pipe', d.get('stage_ms_pipelined'), d.get('planned_frames'))"; }
for rep in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "orbit"; done
KT_NO_PLAN=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit noplan"
python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab"
for rep in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | line "driver-style"; done
