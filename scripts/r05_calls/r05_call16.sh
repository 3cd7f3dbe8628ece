# round 5, call 16b: the level form only while the tracker is alone: the tracker / shell modules, the two-tracker script, then the whole suite twice
cd $GRAFT_REPO_ROOT
python scripts/multistream_one_gpu.py 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -2
python -m pytest tests/test_gpu_tracker.py -m gpu -x -q -k "level or icp" 2>&1 | grep -E "passed|failed|skipped|^E " | tail -3
: > gpurun_out/r05_suite_repeats_final2.log
for i in 1 2; do
echo "=== final tree (level form only while the tracker is alone), run $i: python -m pytest tests/ -x -q -m gpu" >> gpurun_out/r05_suite_repeats_final2.log
python -m pytest tests/ -x -q -m gpu -rs 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | grep -E "passed|failed|SKIPPED|^FAILED|^E " | tail -5 >> gpurun_out/r05_suite_repeats_final2.log
done
cat gpurun_out/r05_suite_repeats_final2.log
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['roofline_stress']; print('fps', round(d['value'],1), 'frac', round(r['frac'],4), 'alone', round(r['frac_alone'],4), d['stage_ms_pipelined'], s['frac_alone'], s['frac_pipelined'])"
