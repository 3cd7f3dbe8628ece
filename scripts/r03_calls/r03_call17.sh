cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### tracker / configs / host shell / pcd / slice tests (download worker)"
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_gpu_host_shell.py tests/test_pcd.py tests/test_slice_process.py tests/test_gpu_volume.py -m gpu -q > gpurun_out/call17_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call17_tests.log | tail -3
echo "#### driver-style x6"
for rep in 1 2 3 4 5 6; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), d['ms_per_step'], 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'ratio', r.get('traffic_ratio'), d['config']['frame_ms'])"; done
echo "#### crabwalk (16 shifts) x2"
for rep in 1 2; do python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['frame_ms'], d['stage_ms_pipelined'])"; done
