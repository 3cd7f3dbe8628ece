cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### track / tracker / configs / host shell tests"
python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_gpu_host_shell.py -m gpu -q > gpurun_out/call11_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call11_tests.log | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],1), 'odo', d['stage_ms']['odometry'], 'pipe', d.get('stage_ms_pipelined'), 'tsdf23 %.1f us alone %.1f frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac']))"; }
echo "#### staged rows A/B (1 = HEAD before the change)"
for rep in 1 2; do
  python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab tree"
  KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab head"
done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit tree"
KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit head"
echo "#### crabwalk kernel stats"
bash scripts/prof_workload.sh crabwalk512 120 r03_crab2 2>&1 | grep -E "joint|residual_kernel|kt_rgb_kernel"
echo "#### pmc traffic"
bash scripts/pmc_traffic.sh orbit512 16 2>&1 | tail -1 | cut -c1-1200
bash scripts/pmc_traffic.sh farwall768 4 2>&1 | tail -1 | cut -c1-1200
