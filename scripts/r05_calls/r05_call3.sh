# round 5, call 3: survey-8c contract tests + report (restructured), ICP squared thresholds (parity + A/B), stress block with parked launches taken out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_tol.py tests/test_gpu_track.py tests/test_golden_ref.py tests/test_gpu_configs.py tests/test_gpu_tracker.py -m gpu -x -q -s > gpurun_out/r05_c3_tests.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r05_c3_tests.log | head
python scripts/tol_contract_report.py > gpurun_out/r05_tol_contract.jsonl 2> gpurun_out/r05_tol_contract.err; cut -c1-420 gpurun_out/r05_tol_contract.jsonl
for i in 1 2; do python bench.py --no-cpu-baseline > gpurun_out/r05_c3_bench_$i.json 2> gpurun_out/r05_c3_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r05_c3_bench_$i.json').read().strip().splitlines()[-1]); r=d['roofline']; s=d['roofline_stress']
print('fps', round(d['value'],1), 'frac', round(r['frac'],4), 'alone', round(r['frac_alone'],4), 'ab', r['contract_ab'], d['stage_ms_pipelined'])
print('stress', {k: s[k] for k in ('frac_alone','frac_pipelined','avg_launch_ms_alone','avg_launch_ms_pipelined','launches_timed_pipelined','survey8c')})"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style fps', round(d['value'],1), 'frac', round(d['roofline']['frac'],4), d['stage_ms_pipelined'])"
