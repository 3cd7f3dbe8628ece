# r04 call 27: final artefacts of the final tree: suite, smoke, bench lines, kernel stats, workloads, probes
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### full GPU suite"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c25_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c25_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/c25_tests.log | head -20
echo "#### smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'tsdf23 %.1f us, alone %.1f, frac %.4f alone %.4f stage_frac %s traffic_ratio %s; pipe %s; stress %.4f ms frac %.4f' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r['frac_alone'], r.get('stage_frac'), r.get('traffic_ratio'), d.get('stage_ms_pipelined'), s.get('avg_launch_ms', 0), s.get('frac', 0)), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; }
echo "#### bench default (with cpu baseline), default x1, driver-style x2"
python bench.py 2>gpurun_out/c25_bench_default.err | tee gpurun_out/r04_bench_default.json | line "default+cpu"
python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04_bench_default_2.json | line "default"
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04_bench_driverstyle_$rep.json | line "driver "; done
echo "#### kernel stats of the default bench"
bash scripts/prof_bench.sh r04_final 2>&1 | tail -3 | cut -c1-200
echo "#### workloads"
bash scripts/run_all_workloads.sh 2>&1 | tail -4
echo "#### crabwalk kernel stats"
bash scripts/prof_workload.sh crabwalk512 120 r04_crabwalk 2>&1 | tail -1 | cut -c1-200
cp "$(find gpurun_out/prof_r04_crabwalk -name '*kernel_stats.csv' | head -1)" gpurun_out/r04_crabwalk512_kernel_stats.csv
echo "#### ICP / joint tail probes"
KT_HIP_LIB=$L/libkt_exp_1.so python scripts/icp_timing.py 2>&1 | tail -7
KT_HIP_LIB=$L/libkt_exp_1.so python scripts/icp_timing.py -ri 2>&1 | head -1
