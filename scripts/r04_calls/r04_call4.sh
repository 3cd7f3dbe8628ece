# r04 call 4: fast-op lean kernel + chained ICP: parity, then A/B (chain 0 / 1) alternating, rates table (more kinds)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### parity: voxel modules + tracker + configs + track"
timeout 1500 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_fullsize.py tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_gpu_host_shell.py -m gpu -q > gpurun_out/c4_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c4_tests.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/c4_tests.log | head -20
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f; serial %s pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], d['stage_ms'], d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)))"; }
echo "#### A/B chain 0 / 1, alternating"
for rep in 1 2 3; do
  for C in 0 1; do
    KT_ICP_CHAIN=$C python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "chain=$C"
  done
done
echo "#### driver-style"
for C in 0 1; do KT_ICP_CHAIN=$C python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | line "driver chain=$C"; done
echo "#### valu rates"
python scripts/valu_rates.py > gpurun_out/r04_valu_rates.md 2> gpurun_out/r04_valu_rates.err; tail -20 gpurun_out/r04_valu_rates.md; tail -2 gpurun_out/r04_valu_rates.err
