# round 5, call 13: the tree with the per-level ICP chain: whole suite, then the measurements that changed -> profiles/r05_*
cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -3
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; python bench.py --no-cpu-baseline > gpurun_out/r05_bench_default_2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_driverstyle_1.json 2>/dev/null; python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_driverstyle_2.json 2>/dev/null
python bench.py --host-frames --no-cpu-baseline --no-stress --no-contract-ab > gpurun_out/r05_bench_hostframes.json 2>/dev/null
: > gpurun_out/r05_workloads.jsonl
for w in orbit256 crabwalk512 farwall768; do python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-contract-ab 2>/dev/null >> gpurun_out/r05_workloads.jsonl; done
for f in r05_bench_default r05_bench_default_2 r05_bench_driverstyle_1 r05_bench_driverstyle_2 r05_bench_hostframes; do python -c "
import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', round(d['value'],1), 'frac', round(r['frac'],4), 'alone', r.get('frac_alone') and round(r['frac_alone'],4), 'traffic_ratio', r.get('traffic_ratio'), d['stage_ms_pipelined'], (d.get('roofline_stress') or {}).get('frac_alone'), (d.get('roofline_stress') or {}).get('frac_pipelined'), (d.get('cpu_baseline') or {}).get('value'))"; done
python -c "
import json
for l in open('gpurun_out/r05_workloads.jsonl'):
    d=json.loads(l); print(d['metric'], round(d['value'],1), d['roofline']['avg_launch_ms'], d['config']['frame_ms']['p50'])"
bash scripts/prof_bench.sh r05_final > /dev/null 2>&1; head -9 gpurun_out/prof_r05_final_kernel_stats.csv | cut -c1-140
