# round 5, call 6: ray-cast batch size A/B (KT_RC_BATCH 4 = tree, 8, 6, 2): parity of the ray-cast tests + bench, alternating
cd $GRAFT_REPO_ROOT
one() {
  echo "== $1"
  KT_HIP_LIB=$2 python -m pytest tests/test_gpu_volume.py tests/test_golden_ref.py -m gpu -x -q -k "raycast or golden" 2>&1 | grep -E "passed|failed" | tail -1
  for i in 1 2; do KT_HIP_LIB=$2 python bench.py --no-cpu-baseline --no-stress --no-contract-ab 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   fps', round(d['value'],1), 'pipelined', d['stage_ms_pipelined'], 'serial raycast', d['stage_ms']['raycast'])"; done
  KT_HIP_LIB=$2 python bench.py --workload farwall768 --steps 24 --warmup 8 --no-cpu-baseline --no-contract-ab 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   farwall768 fps', round(d['value'],1), 'serial raycast', d['stage_ms']['raycast'])"
}
one "tree (batch 4)" ""
one "batch 8" $PWD/exp/libkt_exp_1.so
one "batch 6" $PWD/exp/libkt_exp_2.so
one "batch 2" $PWD/exp/libkt_exp_3.so
one "tree again" ""
