# round 5, call 14: register budget of kt_icp_level_kernel (128 VGPRs = tree; 64 / 80 / 96 with spills): does leaving room for the side streams pay?
cd $GRAFT_REPO_ROOT
one() {
  for i in 1 2; do KT_HIP_LIB=$2 python bench.py --no-cpu-baseline --no-stress --no-contract-ab 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 fps', round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'pipelined', d['stage_ms_pipelined'], 'serial odo', d['stage_ms']['odometry'])"; done
  KT_HIP_LIB=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-contract-ab 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 driver-style fps', round(d['value'],1), 'frac', round(d['roofline']['frac'],4))"
}
one "128" ""
one "64 " $PWD/exp/libkt_exp_1.so
one "80 " $PWD/exp/libkt_exp_2.so
one "96 " $PWD/exp/libkt_exp_3.so
KT_HIP_LIB=$PWD/exp/libkt_exp_1.so python -m pytest tests/test_gpu_tracker.py -m gpu -x -q -k "icp or orbit" 2>&1 | grep -E "passed|failed" | tail -1
