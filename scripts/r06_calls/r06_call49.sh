#!/bin/bash
# the set-up in the odometry launch's epilogue (KT_ICP_FUSED_SETUP=1) once more, on the final schedule: A B A B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c49; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do for m in 0 1; do
  KT_ICP_FUSED_SETUP=$m timeout 600 python bench.py --no-cpu-baseline --no-stress --no-contract-ab > $O/bench_fs${m}_$rep.json 2> $O/bench_fs${m}_$rep.err
  python -c "
import json; j=json.loads(open('$O/bench_fs${m}_$rep.json').read().strip().splitlines()[-1]); r=j['roofline']; print('fused=$m rep $rep fps %.0f' % j['value'], 'frac %.3f alone %.3f' % (r['frac'], r['frac_alone'] or 0), j.get('stage_ms_pipelined'))"
done; done
