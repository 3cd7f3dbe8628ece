# r04 call 1: parity of the lean voxel kernel, the issue-rate calibration (clock under load, SALU model), A/B of the two voxel kernels,
# SQ / GRBM / TCC counters of both on orbit512 and farwall768.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### parity: voxel-kernel modules (both kernels per test)"
timeout 900 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/c1_tests_voxel.log 2>&1; grep -E "passed|failed|error" gpurun_out/c1_tests_voxel.log | tail -3
echo "#### parity: tracker / configs with the lean kernel"
KT_TSDF_LEAN=1 timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_tracker.py -m gpu -q -x > gpurun_out/c1_tests_tracker_lean.log 2>&1; grep -E "passed|failed|error" gpurun_out/c1_tests_tracker_lean.log | tail -3
echo "#### valu rates"
python scripts/valu_rates.py > gpurun_out/r04_valu_rates.md 2> gpurun_out/r04_valu_rates.err; cat gpurun_out/r04_valu_rates.md; tail -2 gpurun_out/r04_valu_rates.err
echo "#### A/B lean 0 / 1"
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', r['kernel'], round(d['value'],1), 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f lane %s; stage_pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], r.get('lane_efficiency'), d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)))"; }
for rep in 1 2; do
  for L in 0 1; do
    KT_TSDF_LEAN=$L python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "lean=$L"
  done
done
echo "#### PMC (orbit512, both kernels)"
bash scripts/pmc_issue.sh orbit512 10 0 orbit_r3 2>&1 | grep -E "PMCI|FAILED"
PASSES="sq sq2 grbm" bash scripts/pmc_issue.sh orbit512 10 1 orbit_lean 2>&1 | grep -E "PMCI|FAILED"
echo "#### PMC (farwall768, both kernels)"
PASSES="sq tcc" bash scripts/pmc_issue.sh farwall768 6 0 far_r3 2>&1 | grep -E "PMCI|FAILED"
PASSES="sq grbm tcc" bash scripts/pmc_issue.sh farwall768 6 1 far_lean 2>&1 | grep -E "PMCI|FAILED"
echo "#### PMC of the micro-benchmark"
bash scripts/pmc_valu_rates.sh 2>&1 | tail -80
echo "#### traffic (lean)"
KT_TSDF_LEAN=1 bash scripts/pmc_traffic.sh farwall768 6 2>&1 | tail -2
