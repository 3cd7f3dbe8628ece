cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### volume / sweep / golden / configs tests (24-bit index arithmetic in the ray cast)"
python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/call14_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call14_tests.log | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],1), 'raycast', d['stage_ms']['raycast'], 'pipe', d.get('stage_ms_pipelined'))"; }
echo "#### ray cast A/B (1 = HEAD)"
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit tree"
  KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit head"
done
for rep in 1 2; do
  python bench.py --workload farwall768 --steps 40 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "farwall tree"
  KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --workload farwall768 --steps 40 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "farwall head"
done
