# r04 call 15: the residual launch without its sweep (every wave of the next launch adds the workgroups' words up itself): parity + A/B on the crab-walk
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### parity"
timeout 1200 python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_golden.py tests/test_gpu_configs.py tests/test_gpu_host_shell.py tests/test_gpu_solve.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E  " | tail -8
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); print('$1', round(d['value'],1), 'p50 %.4f' % d['config']['frame_ms']['p50'], 'pipe', d.get('stage_ms_pipelined'), 'serial', d.get('stage_ms'))"; }
for rep in 1 2; do
  KT_HIP_LIB=$L/libkt_exp_base.so python bench.py --workload crabwalk512 --steps 200 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "base crabwalk"
  python bench.py --workload crabwalk512 --steps 200 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "new  crabwalk"
done
python bench.py --no-cpu-baseline --no-stress 2>/dev/null | line "new default"
