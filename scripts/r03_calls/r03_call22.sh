cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### full GPU suite"
python -m pytest tests -m gpu -q > gpurun_out/call22_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call16_tests.log | tail -3
echo "#### smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "#### default bench (driver style, then the long default)"
for rep in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_driverstyle_$rep.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_driverstyle_$rep.json')); print(round(d['value'],1), d['ms_per_step'], 'frac', round(d['roofline']['frac'],4), 'ratio', d['roofline'].get('traffic_ratio'), 'stress', round((d.get('roofline_stress') or {}).get('frac',0),4), 'cpu', d['cpu_baseline']['value'], d['planned_frames'])"; done
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_default_final.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default_final.json')); print(json.dumps({k: d[k] for k in d if k not in ('config',)})[:3000])"
echo "#### all workloads"
bash scripts/run_all_workloads.sh 2>&1 | tail -8
echo "#### kernel stats"
bash scripts/prof_bench.sh r03_final 2>&1 | tail -26
echo "#### pmc traffic"
bash scripts/pmc_traffic.sh orbit512 16 2>&1 | tail -1 | cut -c1-1500
bash scripts/pmc_traffic.sh farwall768 4 2>&1 | tail -1 | cut -c1-1500
