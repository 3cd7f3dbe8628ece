cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### volume / sweep / golden / track / tracker / configs tests (task list: counts in the interval kernel, one-workgroup scan staged in LDS, wave-per-pair placement)"
python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/call19_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call19_tests.log | tail -3
KT_TSDF_WCX=32 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'serial odo', d['stage_ms']['odometry'], 'integ', d['stage_ms']['integrate'], 'tsdf23 %.1f us alone %.1f frac %.4f stress %.4f ms' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], s.get('avg_launch_ms', 0)), 'pipe', d.get('stage_ms_pipelined'))"; }
for rep in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "orbit"; done
KT_NO_PLAN=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit noplan"
python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab"
echo "#### kernel stats (no stress leg)"
bash scripts/prof_bench.sh r03_final 2>&1 | tail -24
