#!/bin/bash
# final tree: rocprofv3 kernel statistics (default bench, farwall768, crabwalk512 -ri), the overlap reports and the stream timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c41; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash scripts/prof_bench.sh r06_final > $O/prof_bench.log 2>&1; tail -3 $O/prof_bench.log | cut -c1-200
cp gpurun_out/prof_r06_final_kernel_stats.csv $O/r06_final_kernel_stats.csv; cp gpurun_out/prof_r06_final_bench.json $O/r06_final_stats_bench.json
bash scripts/prof_workload.sh farwall768 40 r06_farwall768 > $O/prof_far.log 2>&1; cp gpurun_out/prof_r06_farwall768_kernel_stats.csv $O/r06_kernel_stats_farwall768.csv
bash scripts/prof_workload.sh crabwalk512 200 r06_crabwalk512 > $O/prof_crab.log 2>&1; cp gpurun_out/prof_r06_crabwalk512_kernel_stats.csv $O/r06_kernel_stats_crabwalk512.csv
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_orbit -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-contract-ab --no-stress > $R/$O/trace_orbit.log 2>&1
T=$(find $R/$O/trace_orbit -name '*kernel_trace.csv' | head -1)
python $R/scripts/stream_timeline.py "$T" > $R/$O/r06_timeline_orbit512.txt 2>&1
python $R/scripts/overlap_report.py "$T" > $R/$O/r06_overlap_orbit512_tsdf23.txt 2>&1
python $R/scripts/overlap_report.py "$T" kt_raycast_kernel > $R/$O/r06_overlap_orbit512_raycast.txt 2>&1
python $R/scripts/overlap_report.py "$T" kt_icp_level_kernel > $R/$O/r06_overlap_orbit512_icp.txt 2>&1
python - "$T" <<'PY' > $R/$O/r06_main_stream_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.Counter(r['Stream_Id'] for r in rows)
main = by.most_common(1)[0][0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','')[:36]) for r in rows if r['Stream_Id'] == main)
idx = [i for i,(s,e,n) in enumerate(ev) if n.startswith('kt_icp_level_kernel')]
idx = idx[len(idx)//5: 4*len(idx)//5]
gaps = collections.defaultdict(float); durs = collections.defaultdict(float); n = 0
for a, b in zip(idx[:-1], idx[1:]):
    if b - a != 4: continue   # icp, setup, tsdf23, raycast
    n += 1
    for k in range(a, b):
        durs[ev[k][2]] += ev[k][1] - ev[k][0]
        gaps['before ' + ev[k+1][2]] += ev[k+1][0] - ev[k][1]
print('main stream, frames of exactly {odometry, set-up, voxel kernel, ray cast}:', n)
for k, v in durs.items(): print('  dur  %-40s %7.2f us' % (k, v / n / 1e3))
for k, v in gaps.items(): print('  gap  %-40s %7.2f us' % (k, v / n / 1e3))
PY
rm -rf $R/$O/trace_orbit $R/gpurun_out/prof_r06_final $R/gpurun_out/prof_r06_farwall768 $R/gpurun_out/prof_r06_crabwalk512
head -12 $R/$O/r06_overlap_orbit512_tsdf23.txt; cat $R/$O/r06_main_stream_gaps.txt; head -20 $R/$O/r06_final_kernel_stats.csv | cut -c1-150
