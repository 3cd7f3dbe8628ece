# r04 call 21: final artefacts of the tree
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### full GPU suite"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c21_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c21_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/c21_tests.log | head -20
echo "#### smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "#### PMC traffic of the voxel kernel (the bench quotes these files)"
bash scripts/pmc_traffic.sh farwall768 6 2>&1 | tail -1 | cut -c1-600
bash scripts/pmc_traffic.sh orbit512 16 2>&1 | tail -1 | cut -c1-600
cp gpurun_out/r04_pmc_tsdf23_farwall768.json gpurun_out/r04_pmc_tsdf23_orbit512.json profiles/
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f stage_frac %s traffic_ratio %s; pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], r.get('stage_frac'), r.get('traffic_ratio'), d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; }
echo "#### bench default (with cpu baseline), default x1, driver-style x2"
python bench.py 2>gpurun_out/c21_bench_default.err | tee gpurun_out/r04_bench_default.json | line "default+cpu"
python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04_bench_default_2.json | line "default"
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04_bench_driverstyle_$rep.json | line "driver "; done
echo "#### kernel stats of the default bench"
bash scripts/prof_bench.sh r04_final 2>&1 | tail -24
echo "#### workloads"
bash scripts/run_all_workloads.sh 2>&1 | tail -6
echo "#### crabwalk kernel stats"
bash scripts/prof_workload.sh crabwalk512 120 r04_crabwalk 2>&1 | tail -3
cp "$(find gpurun_out/prof_r04_crabwalk -name '*kernel_stats.csv' | head -1)" gpurun_out/r04_crabwalk512_kernel_stats.csv
echo "#### slice stage"
bash scripts/slice_profile.sh 2>&1 | tail -4
echo "#### ICP tail probes"
KT_HIP_LIB=$L/libkt_exp_1.so python scripts/icp_timing.py 2>&1 | tail -1
echo "#### issue counters of the voxel kernel"
PASSES="sq sq2" bash scripts/pmc_issue.sh orbit512 10 1 orbit_final 2>&1 | grep -E "PMCI|FAILED" | grep tsdf23 | cut -c1-500
PASSES="sq tcc" bash scripts/pmc_issue.sh farwall768 6 1 far_final 2>&1 | grep -E "PMCI|FAILED" | grep tsdf23 | cut -c1-500
