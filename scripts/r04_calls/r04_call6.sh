# r04 call 6: two-kernel task list, windowed kNN fallback of the slice stage: parity, timing, kernel stats
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### parity"
timeout 1500 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_fullsize.py tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_slice_process.py tests/test_pcd.py -m gpu -q > gpurun_out/c6_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c6_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/c6_tests.log | head -20
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f; serial %s pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], d['stage_ms'], d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)))"; }
echo "#### A/B task list serial (r3) / two-kernel, alternating"
for rep in 1 2; do
  KT_TSDF_TASKS_SERIAL=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "tasks=serial"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "tasks=grid  "
done
echo "#### slice stage"
python scripts/slice_stage_timing.py 2>/dev/null | tail -4
echo "#### kernel stats (default bench)"
bash scripts/prof_bench.sh r04_c6 2>&1 | tail -26
