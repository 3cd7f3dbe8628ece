# r04 call 29: the first pixel of every thread requested before the pose (the pose is a round trip of its own at the head of every iteration): parity, probes, A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### parity"
timeout 1200 python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_golden.py tests/test_gpu_configs.py tests/test_gpu_host_shell.py tests/test_gpu_solve.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E  " | tail -8
echo "#### probes"
KT_HIP_LIB=$L/libkt_exp_1.so python scripts/icp_timing.py 2>&1 | head -1
KT_HIP_LIB=$L/libkt_exp_1.so python scripts/icp_timing.py -ri 2>&1 | head -1
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'pipe', d.get('stage_ms_pipelined'), 'serial', d.get('stage_ms'))"; }
for rep in 1 2; do
  KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "base  "
  python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "prefet"
done
KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --workload crabwalk512 --steps 200 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "base   crabwalk"
python bench.py --workload crabwalk512 --steps 200 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "prefet crabwalk"
