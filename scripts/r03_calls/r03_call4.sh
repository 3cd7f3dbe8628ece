cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### pytest -m gpu"; python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_call4.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03_pytest_call4.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/r03_pytest_call4.log | head -20
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['stage_ms'], d.get('stage_ms_pipelined'), d.get('planned_frames'), 'frac', round(d['roofline']['frac'],4), 'tsdf23 in-frame ms', round(d['roofline']['avg_launch_ms'],4), d['host_ms_per_frame'], d['config']['frame_ms'])"; }
for rep in 1 2; do
  echo "#### bench planned $rep"; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | tee gpurun_out/r03_bench_plan2_$rep.json | line plan
  echo "#### bench KT_NO_PLAN $rep"; KT_NO_PLAN=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | tee gpurun_out/r03_bench_noplan2_$rep.json | line noplan
done
echo "#### crabwalk"; python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | tee gpurun_out/r03_bench_crabwalk2.json | line crab
echo "#### variants"; KT_SKIP_TREE=1 bash scripts/variants_ab.sh 40 2>&1 | tee gpurun_out/r03_variants_call4.log
echo "#### pmc"; bash scripts/pmc_variants.sh 2>&1 | grep PMC | tee gpurun_out/r03_pmc_variants_call4.log
echo "#### kernel trace"; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03b -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress > $GRAFT_REPO_ROOT/gpurun_out/prof_r03b.log 2>&1); f=$(ls -t gpurun_out/prof_r03b/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r03_kernel_stats_plan2.csv; head -12 $f | cut -c1-140
python scripts/frame_timeline.py $(ls -t gpurun_out/prof_r03b/*/*kernel_trace.csv | head -1)
