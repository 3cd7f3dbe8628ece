#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c34; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && KT_BENCH_PERIODS=1 rocprofv3 --kernel-trace -d $R/$O/trace -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-contract-ab > $R/$O/bench.json 2> $R/$O/bench.err )
python - <<'PY'
import sqlite3, json
j=json.loads(open("gpurun_out/c34/bench.json").read().strip().splitlines()[-1]); print("fps %.0f" % j["value"], j["config"].get("periods_ms"))
db=sqlite3.connect("gpurun_out/c34/trace/p_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; sym=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=[(a,b,q,n) for a,b,q,n in cur.execute(f"select k.start,k.end,k.queue_id,s.kernel_name from {kd} k join {sym} s on k.kernel_id=s.id order by k.start")]
icp=[i for i,r in enumerate(rows) if 'kt_icp_level_kernel' in r[3]]
print("icp launches", len(icp))
t0=rows[icp[0]][0]
for n,i in enumerate(icp[:34]):
    s,e,q,_=rows[i]
    nxt=rows[icp[n+1]][0] if n+1<len(icp) else e
    seg=[r for r in rows if s<=r[0]<nxt and r[2]==q]
    desc=" ".join("%s %.0f" % (r[3].split("kt_")[1][:10] if "kt_" in r[3] else r[3][:10], (r[1]-r[0])/1e3) for r in seg)
    print("icp#%d at %.0f us, frame %.0f us: %s" % (n, (s-t0)/1e3, (nxt-s)/1e3, desc))
    if any("extract" in r[3] for r in seg):
        allr=[r for r in rows if s<=r[0]<nxt]
        for r in allr: print("      +%.1f .. +%.1f  q%s %s" % ((r[0]-s)/1e3, (r[1]-s)/1e3, r[2], r[3][:40]))
PY
rm -rf $R/$O/trace
