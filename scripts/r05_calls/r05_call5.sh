# round 5, call 5: bilateral v2 (sentinel border), raycast hit processing per axis, tile kernels fused -- parity + timings
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_image.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r05_c5_tests.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r05_c5_tests.log | head
python scripts/bilateral_ab.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_gpu_configs.py tests/test_gpu_tracker.py -m gpu -x -q > gpurun_out/r05_c5_tests2.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r05_c5_tests2.log | head
for i in 1 2; do python bench.py --no-cpu-baseline --no-stress | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('fps', round(d['value'],1), 'frac', round(r['frac'],4), 'alone', round(r['frac_alone'],4), d['stage_ms_pipelined'], d['stage_ms'])"; done
python bench.py --workload farwall768 --steps 40 --warmup 10 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('farwall768 fps', round(d['value'],1), 'tsdf23 in region', d['roofline']['avg_launch_ms'], d['stage_ms_pipelined'], d['stage_ms'])"
python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('crabwalk512 fps', round(d['value'],1), d['stage_ms_pipelined'])"
