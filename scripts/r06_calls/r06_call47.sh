#!/bin/bash
# dense view: the level form of the odometry (KT_DENSE_STEPWISE=0) against the stepwise default, after the read-ahead and plan changes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c47; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for m in 1 0; do
  KT_DENSE_STEPWISE=$m timeout 600 python bench.py --workload farwall768 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --no-contract-ab > $O/far_sw${m}_$rep.json 2> $O/far_sw${m}_$rep.err
  python -c "
import json; j=json.loads(open('$O/far_sw${m}_$rep.json').read().strip().splitlines()[-1]); r=j['roofline']; print('stepwise=$m rep $rep fps %.1f' % j['value'], 'frac %.3f alone %.3f' % (r['frac'], r['frac_alone'] or 0), j.get('stage_ms_pipelined'), j.get('planned_frames'), 'fallbacks', j['config'].get('odometry_fallbacks'))"
done; done
