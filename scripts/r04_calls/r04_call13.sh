# r04 call 13: ray-cast march step on the fast instruction class (magic-add floor, launch constants in VGPRs, buffer loads): parity + A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### parity"
timeout 1200 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_tracker.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E  " | tail -8
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'pipe', d.get('stage_ms_pipelined'), 'serial', d.get('stage_ms'))"; }
for rep in 1 2; do
  KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "base    "
  python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "new rc  "
done
KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --workload farwall768 --steps 40 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "base farwall"
python bench.py --workload farwall768 --steps 40 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "new  farwall"
