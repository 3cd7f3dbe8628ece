cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### volume / sweep / golden / track / tracker tests (record gathers through a descriptor, 24-bit index in the ICP row)"
python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_track.py tests/test_gpu_tracker.py -m gpu -q > gpurun_out/call15_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call15_tests.log | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'odo', d['stage_ms']['odometry'], 'tsdf23 %.1f us alone %.1f frac %.4f stress %.4f ms' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], s.get('avg_launch_ms', 0)), 'pipe', d.get('stage_ms_pipelined'))"; }
echo "#### A/B (1 = without)"
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "orbit tree "
  KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "orbit plain"
done
