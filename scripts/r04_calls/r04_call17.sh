# r04 call 17: the voxel kernel's start-up in three dependent round trips instead of five: parity, timeline, A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### parity"
timeout 1200 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_tracker.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E  " | tail -8
echo "#### timeline"
KT_HIP_LIB=$L/libkt_exp_1.so python scripts/tsdf_timeline.py orbit512 2>&1 | tail -14
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f stage_frac %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], r.get('stage_frac'), s.get('avg_launch_ms', 0), s.get('frac', 0)), 'pipe', d.get('stage_ms_pipelined'))"; }
for rep in 1 2; do
  KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --no-cpu-baseline 2>/dev/null | line "base   "
  python bench.py --no-cpu-baseline 2>/dev/null | line "startup"
done
KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | line "base    driver"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | line "startup driver"
