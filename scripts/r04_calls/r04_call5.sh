# r04 call 5: chained ICP with cached granule loads, lean kernel constants in VGPRs, f2 exact normals + its kernel profile
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### parity: track / tracker / slice / pcd / volume"
timeout 1500 python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_slice_process.py tests/test_pcd.py tests/test_gpu_volume.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/c5_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c5_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/c5_tests.log | head -20
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f; serial odo %.4f pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], d['stage_ms']['odometry'], d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)))"; }
echo "#### A/B chain 0 / 1, alternating"
for rep in 1 2; do
  for C in 0 1; do
    KT_ICP_CHAIN=$C python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "chain=$C"
  done
done
echo "#### slice stage profile"
bash scripts/slice_profile.sh 2>&1 | tail -45
