#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c16; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_orbit -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-contract-ab --no-stress > $R/$O/trace_orbit.log 2>&1
T=$(find $R/$O/trace_orbit -name '*kernel_trace.csv' | head -1)
python $R/scripts/stream_timeline.py "$T" | tee $R/$O/stream_timeline_orbit.txt
python - "$T" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.Counter(r['Stream_Id'] for r in rows)
main = by.most_common(1)[0][0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','')[:36]) for r in rows if r['Stream_Id'] == main)
idx = [i for i,(s,e,n) in enumerate(ev) if n.startswith('kt_icp_level_kernel')]
idx = idx[len(idx)//5: 2*len(idx)//5]
gaps = collections.defaultdict(float); durs = collections.defaultdict(float); n = 0
for a, b in zip(idx[:-1], idx[1:]):
    if b - a != 4: continue   # icp, setup, tsdf23, raycast
    n += 1
    for k in range(a, b):
        durs[ev[k][2]] += ev[k][1] - ev[k][0]
        gaps['before ' + ev[k+1][2]] += ev[k+1][0] - ev[k][1]
print('frames', n)
for k, v in durs.items(): print('  dur  %-40s %7.2f us' % (k, v / n / 1e3))
for k, v in gaps.items(): print('  gap  %-40s %7.2f us' % (k, v / n / 1e3))
PY
rm -rf $R/$O/trace_orbit
