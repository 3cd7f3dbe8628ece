# r04 call 3: the whole GPU suite (no -x), then lean / r3 voxel kernel alternating on one box, three times
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c3_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c3_tests.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/c3_tests.log | head -20
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', r['kernel'], round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f; pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)))"; }
echo "#### A/B lean 0 / 1, alternating"
for rep in 1 2 3; do
  for L in 0 1; do
    KT_TSDF_LEAN=$L python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "lean=$L"
  done
done
