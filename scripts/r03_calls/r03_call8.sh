cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### pytest -m gpu"; python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_call8.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03_pytest_call8.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/r03_pytest_call8.log | head -20
echo "#### smoke"; python __graft_entry__.py --smoke 2>&1 | tail -2
echo "#### slice timing"; python scripts/slice_stage_timing.py 2>/dev/null | tee gpurun_out/r03_slice_stage_timing.md | grep "^|"
echo "#### bench default"; python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']; s=d['roofline_stress']
print(round(d['value'],1), d['stage_ms'], d['stage_ms_pipelined'], d['planned_frames'], 'frac', round(r['frac'],4), round(r['frac_alone'],4), r['lane_efficiency'], 'stress', round(s['frac'],4), s['avg_launch_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['pose_max_abs_diff_vs_gpu'])"
echo "#### bench driver-style"; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['planned_frames'])"
echo "#### pmc traffic"; for w in orbit512 farwall768; do S=16; [ $w = farwall768 ] && S=6; bash scripts/pmc_traffic.sh $w $S 2>&1 | tail -1 | cut -c1-600; done
