# r04 call 18: waves per workgroup of the voxel kernel (dispatch ramp, table fill shared by more waves): timeline by entry quartile, A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### timeline, 4 waves per workgroup"
KT_HIP_LIB=$L/libkt_exp_1.so python scripts/tsdf_timeline.py orbit512 2>&1 | tail -18
echo "#### timeline, 16 waves per workgroup"
KT_TL_WPB=16 KT_HIP_LIB=$L/libkt_exp_4.so python scripts/tsdf_timeline.py orbit512 2>&1 | tail -18
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], s.get('avg_launch_ms', 0), s.get('frac', 0)))"; }
python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 4 "
KT_HIP_LIB=$L/libkt_exp_5.so python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 2 "
KT_HIP_LIB=$L/libkt_exp_2.so python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 8 "
KT_HIP_LIB=$L/libkt_exp_3.so python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 16"
python bench.py --no-cpu-baseline 2>/dev/null | line "wpb 4 "
