# round 5, call 2: the survey-8c contract (tests + counted report), bench with the new blocks, farwall768 kernel trace, host-frames line
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_tol.py tests/test_gpu_track.py -m gpu -x -q -s > gpurun_out/r05_c2_tests.log 2>&1; tail -4 gpurun_out/r05_c2_tests.log
python scripts/tol_contract_report.py > gpurun_out/r05_tol_contract.jsonl 2> gpurun_out/r05_tol_contract.err; cat gpurun_out/r05_tol_contract.jsonl | cut -c1-700
python bench.py > gpurun_out/r05_c2_bench.json 2> gpurun_out/r05_c2_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r05_c2_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; s=d['roofline_stress']
print('fps', d['value'], 'frac', r['frac'], 'alone', r['frac_alone'], 'ab', r['contract_ab'])
print('stress', {k: s[k] for k in ('frac_alone','frac_pipelined','avg_launch_ms_alone','avg_launch_ms_pipelined','survey8c')})"
python bench.py --host-frames --no-cpu-baseline --no-stress > gpurun_out/r05_bench_hostframes.json 2>/dev/null; cut -c1-200 gpurun_out/r05_bench_hostframes.json
bash scripts/prof_workload.sh farwall768 40 r05_farwall768
