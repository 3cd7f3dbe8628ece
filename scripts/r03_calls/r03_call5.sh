cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### pytest -m gpu"; python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_call5.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03_pytest_call5.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/r03_pytest_call5.log | head -20
echo "#### slice stage timing"; python scripts/slice_stage_timing.py 2>&1 | tee gpurun_out/r03_slice_stage_timing.md | tail -12
echo "#### bench"; python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r03_bench_call5.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms'], d.get('stage_ms_pipelined'), d.get('planned_frames'), 'frac', round(d['roofline']['frac'],4), round(d['roofline']['frac_alone'],4), 'stress', d['roofline_stress']['frac'], d['roofline_stress']['avg_launch_ms'])"
