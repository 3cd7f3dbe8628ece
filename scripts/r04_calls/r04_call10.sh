# r04 call 10: is the serial tail of the ICP kernel instruction-fetch bound?  (KT_ICP_TWICE: the tail run 2-3 times by the same thread; KT_TAIL_WARM: every workgroup walks it)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### tail timing, tree code (variant 1)"; KT_HIP_LIB=$L/libkt_exp_1.so python scripts/icp_timing.py 2>&1 | tail -1
echo "#### tail timing, warm tail (variant 3)"; KT_HIP_LIB=$L/libkt_exp_3.so python scripts/icp_timing.py 2>&1 | tail -1
echo "#### parity with the warm tail (variant 2)"
KT_HIP_LIB=$L/libkt_exp_2.so timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_golden.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'pipe', d.get('stage_ms_pipelined'), 'serial', d.get('stage_ms'))"; }
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "tree  "
  KT_HIP_LIB=$L/libkt_exp_2.so python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "warm  "
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | line "tree driver"
KT_HIP_LIB=$L/libkt_exp_2.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | line "warm driver"
