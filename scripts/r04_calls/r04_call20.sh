# r04 call 20: longer tasks (32 z) in batches of 8 z-steps on fewer waves (4 / 6 per SIMD): parity, A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### parity, variant 1"
KT_HIP_LIB=$L/libkt_exp_1.so timeout 900 python -m pytest tests/test_gpu_volume.py tests/test_golden.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E  " | tail -6
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f lane_eff %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], r.get('lane_efficiency'), s.get('avg_launch_ms', 0), s.get('frac', 0)), 'err', d['config']['pose_err_m_at_end'])"; }
python bench.py --no-cpu-baseline 2>/dev/null | line "tree          "
KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --no-cpu-baseline 2>/dev/null | line "32z/8, 4096 w "
KT_HIP_LIB=$L/libkt_exp_2.so python bench.py --no-cpu-baseline 2>/dev/null | line "32z/8, 8192 w4"
KT_HIP_LIB=$L/libkt_exp_3.so python bench.py --no-cpu-baseline 2>/dev/null | line "32z/8, 6144 w "
python bench.py --no-cpu-baseline 2>/dev/null | line "tree          "
