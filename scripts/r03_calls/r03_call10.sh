cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### track / tracker / configs tests with the LDS-system epilogue"
python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/call10_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call10_tests.log | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],1), 'odo', d['stage_ms']['odometry'], 'pipe', d.get('stage_ms_pipelined'), 'tsdf23 %.1f us alone %.1f frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac']))"; }
echo "#### epilogue A/B (6 = old)"
for rep in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit tree"
  KT_HIP_LIB=$L/libkt_exp_6.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit old "
  python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab tree"
  KT_HIP_LIB=$L/libkt_exp_6.so python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab old "
done
echo "#### balanced dealing: parity"
KT_HIP_LIB=$L/libkt_exp_1.so python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -1
KT_HIP_LIB=$L/libkt_exp_1.so python -m pytest tests/test_gpu_tracker.py -m gpu -q -k "planned or readahead or shift" 2>&1 | grep -E "passed|failed|error" | tail -1
echo "#### tsdf23 variants (1 balanced, 2 nt ld+st, 3 nt st, 4 nt ld)"
for rep in 1 2; do
for i in 0 1 2 3 4; do
  lib=""; [ $i != 0 ] && lib=$L/libkt_exp_$i.so
  KT_HIP_LIB=$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('roofline_stress') or {}
print('v$i', round(d['value'],1), 'tsdf23 %.1f us, alone %.1f, frac %.4f; raycast %.4f; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], d['stage_ms']['raycast'], s.get('avg_launch_ms', 0), s.get('frac', 0)))"
done; done
echo "#### icp timing (variant 5)"
KT_HIP_LIB=$L/libkt_exp_5.so python scripts/icp_timing.py 2>&1 | tail -2
echo "#### crabwalk kernel stats"
bash scripts/prof_workload.sh crabwalk512 120 r03_crab 2>&1 | tail -26
