# r04 call 19: 2 waves per workgroup: parity first, then timeline and A/B with the stage times
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### parity, 2 waves per workgroup"
KT_HIP_LIB=$L/libkt_exp_1.so timeout 1200 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_configs.py tests/test_gpu_tracker.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E  " | tail -8
echo "#### timeline, 2 waves per workgroup"
KT_TL_WPB=2 KT_HIP_LIB=$L/libkt_exp_2.so python scripts/tsdf_timeline.py orbit512 2>&1 | tail -18
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f U %.0f; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], r['U_voxels_updated'], s.get('avg_launch_ms', 0), s.get('frac', 0)), 'err', d['config']['pose_err_m_at_end'], 'pipe', d.get('stage_ms_pipelined'), 'serial', d.get('stage_ms'))"; }
python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 4 "
KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 2 "
KT_HIP_LIB=$L/libkt_exp_3.so python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 1 "
python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 4 "
KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 2 "
