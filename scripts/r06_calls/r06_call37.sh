#!/bin/bash
# dense view (farwall768) under both announce orders: frame rate, then a kernel trace of each (two frames printed)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c37; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in first behind; do
  if [ $m = first ]; then export KT_BENCH_ANNOUNCE_FIRST=1; else unset KT_BENCH_ANNOUNCE_FIRST; fi
  for rep in 1 2; do
    timeout 600 python bench.py --workload farwall768 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --no-contract-ab > $O/far_${m}_$rep.json 2> $O/far_${m}_$rep.err
    python -c "
import json; j=json.loads(open('$O/far_${m}_$rep.json').read().strip().splitlines()[-1]); print('$m $rep fps %.1f' % j['value'], j['roofline']['frac'], j.get('stage_ms_pipelined'), j.get('planned_frames'), j['config'].get('side_gate'))"
  done
  ( cd /tmp && rocprofv3 --kernel-trace -d $R/$O/trace_$m -o p -- python $R/bench.py --workload farwall768 --steps 12 --warmup 6 --no-cpu-baseline --no-stress --no-contract-ab > $R/$O/trace_$m.json 2> $R/$O/trace_$m.err )
  python - <<PY
import sqlite3
db=sqlite3.connect("$O/trace_$m/p_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; sym=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=[(a,b,q,n) for a,b,q,n in cur.execute(f"select k.start,k.end,k.queue_id,s.kernel_name from {kd} k join {sym} s on k.kernel_id=s.id order by k.start")]
ts=[i for i,r in enumerate(rows) if 'tsdf23_lean_kernelILb0' in r[3]]
print("$m: voxel launches", len(ts))
a=rows[ts[9]][0]; b=rows[ts[11]][0]
mainq=rows[ts[9]][2]
prev=None
for r in rows:
    if a<=r[0]<b:
        nm=r[3].split('kt_')[1][:22] if 'kt_' in r[3] else r[3][:22]
        if r[2]==mainq and ('icp_kernel' in r[3]):
            if prev is None: prev=[r[0],r[1],1]
            else: prev[1]=r[1]; prev[2]+=1
            continue
        if prev: print("   q%s +%7.1f .. +%7.1f  icp x%d" % (mainq,(prev[0]-a)/1e3,(prev[1]-a)/1e3,prev[2])); prev=None
        print("   q%s +%7.1f .. +%7.1f  %s" % (r[2],(r[0]-a)/1e3,(r[1]-a)/1e3,nm))
PY
  rm -rf $R/$O/trace_$m
done
