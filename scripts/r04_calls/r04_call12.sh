# r04 call 12: pivot order by ranks (the swap emulation only on ties): both tail forms on the same systems, parity, timing, A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### both forms on the same systems"
timeout 600 python -m pytest tests/test_gpu_solve.py -m gpu -x -q 2>&1 | tail -4
echo "#### trajectory parity"
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_golden.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E  " | tail -8
echo "#### tail timing (10 ns ticks)"; KT_HIP_LIB=$L/libkt_exp_1.so python scripts/icp_timing.py 2>&1 | tail -1
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'pipe', d.get('stage_ms_pipelined'), 'serial', d.get('stage_ms'))"; }
for rep in 1 2; do
  KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "serial tail"
  python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "lane tail  "
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | line "lane tail driver-style"
python bench.py --workload crabwalk512 --steps 200 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "lane tail   crabwalk"
