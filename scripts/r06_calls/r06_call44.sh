#!/bin/bash
# final tree: bench lines, the workloads table, rocprofv3 kernel statistics, overlap reports and the main-stream gaps (the PMC traffic of call 40 still holds: kt_volume.hip is unchanged)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c44; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err; echo "default rc $?"
timeout 900 python bench.py --no-cpu-baseline > $O/r06_bench_default_2.json 2> $O/bench_default2.err; echo "default2 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driverstyle_1.json 2> $O/bench_drv1.err; echo "drv1 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_driverstyle_2.json 2> $O/bench_drv2.err; echo "drv2 rc $?"
timeout 900 python bench.py --host-frames --no-cpu-baseline --no-stress > $O/r06_bench_hostframes.json 2> $O/bench_host.err; echo "host rc $?"
bash scripts/run_all_workloads.sh > $O/workloads.log 2>&1; cp gpurun_out/workloads_r06.jsonl $O/r06_workloads.jsonl; tail -5 $O/workloads.log | cut -c1-200
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c44/r06_bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "lane", r.get("lane_efficiency"), "traffic_ratio", r.get("traffic_ratio"), "serial", j.get("stage_ms"), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / %.3f sol %s tr %s" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0, (s.get("speed_of_light") or {}).get("frac_alone"), s.get("traffic_ratio")), "cpu", (j.get("cpu_baseline") or {}).get("value"))
PY
O=gpurun_out/c44; mkdir -p $O
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
