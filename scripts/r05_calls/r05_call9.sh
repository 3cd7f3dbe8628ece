# round 5, call 9: the overlapped ICP chain at the bench's size: A/B of the frame rate, then parity
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do KT_ICP_OVERLAP=$v timeout 120 python bench.py --no-cpu-baseline --no-stress --no-contract-ab 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $v fps', round(d['value'],1), 'pipelined', d['stage_ms_pipelined'], 'serial', d['stage_ms']['odometry'], 'err', d['config']['pose_err_m_at_end'])"; done
KT_ICP_OVERLAP=1 timeout 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_tracker.py tests/test_gpu_track.py -m gpu -x -q > gpurun_out/r05_c9_tests.log 2>&1; grep -E "passed|failed|^E |Timeout" gpurun_out/r05_c9_tests.log | head
