# round 5, call 12: the ICP chain as one launch per level (kt_icp_level_kernel): parity, then A/B of the frame rate against one launch per iteration
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_track.py -m gpu -x -q -k "icp or handoff or orbit" > gpurun_out/r05_c12_tests.log 2>&1; grep -E "passed|failed|^E |Timeout" gpurun_out/r05_c12_tests.log | head
for v in 0 1 0 1; do KT_ICP_LEVELS=$v timeout 120 python bench.py --no-cpu-baseline --no-stress --no-contract-ab 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('levels $v fps', round(d['value'],1), 'pipelined', d['stage_ms_pipelined'], 'serial', d['stage_ms']['odometry'], 'err', d['config']['pose_err_m_at_end'])"; done
timeout 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_tracker.py tests/test_golden.py -m gpu -x -q > gpurun_out/r05_c12_tests2.log 2>&1; grep -E "passed|failed|^E |Timeout" gpurun_out/r05_c12_tests2.log | head
