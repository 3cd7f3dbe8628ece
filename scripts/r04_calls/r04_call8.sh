# r04 call 8: NT launch property parity, bench lines, PMC (issue + traffic) of the final voxel kernel, kernel stats
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### parity: voxel modules + configs"
timeout 1500 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/c8_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c8_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/c8_tests.log | head -20
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f stage_frac %s; pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], r.get('stage_frac'), d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)))"; }
echo "#### bench default x2, driver-style x2"
for rep in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | line "default"; done
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver "; done
echo "#### PMC issue counters"
PASSES="sq sq2" bash scripts/pmc_issue.sh orbit512 10 1 orbit_final 2>&1 | grep -E "PMCI|FAILED" | grep tsdf23
PASSES="sq tcc" bash scripts/pmc_issue.sh farwall768 6 1 far_final 2>&1 | grep -E "PMCI|FAILED" | grep tsdf23
echo "#### traffic"
bash scripts/pmc_traffic.sh farwall768 6 2>&1 | tail -1 | cut -c1-900
bash scripts/pmc_traffic.sh orbit512 16 2>&1 | tail -1 | cut -c1-900
