cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### volume / sweep / golden / tracker / configs tests (run-time wave-column shape, verified divisions in the ray cast)"
python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_tracker.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/call13_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call13_tests.log | tail -3
echo "#### same volume tests with the other shape forced"
KT_TSDF_WCX=32 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -1
KT_TSDF_WCX=16 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],1), 'raycast', d['stage_ms']['raycast'], 'pipe', d.get('stage_ms_pipelined'), 'tsdf23 %.1f us alone %.1f frac %.4f lane %s' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r.get('lane_efficiency')))"; }
echo "#### ray cast divisions A/B"
for rep in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit fastdiv"
  KT_NO_FASTDIV=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit ieee   "
done
echo "#### wave-column shape per workload"
for w in orbit512 orbit256 crabwalk512 farwall768; do
  st=200; [ $w = farwall768 ] && st=40
  for rep in 1 2; do
    KT_TSDF_WCX=32 python bench.py --workload $w --steps $st --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "$w 32x2"
    KT_TSDF_WCX=16 python bench.py --workload $w --steps $st --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "$w 16x4"
  done
done
