#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c4; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-contract-ab --no-stress > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
for rep in a b; do
  run cur_$rep KT_X=1
  run noexit_$rep KT_HIP_LIB=$R/variants/lvl_noexit.so
  run twosets_$rep KT_HIP_LIB=$R/variants/lvl_noexit_twosets.so
  (cd r05tree && timeout 600 python bench.py --no-cpu-baseline --no-contract-ab --no-stress > ../$O/bench_r05_$rep.json 2> ../$O/bench_r05_$rep.err; echo "r05 rc $?")
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c4/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "odo_pipe", (j.get("stage_ms_pipelined") or {}).get("odometry"), "odo_serial", (j.get("stage_ms") or {}).get("odometry"))
PY
# what runs beside the voxel kernel on the dense view, pipelined
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_far -- python $R/bench.py --workload farwall768 --steps 12 --warmup 8 --no-cpu-baseline --no-contract-ab > $R/$O/trace_far.log 2>&1
T=$(find $R/$O/trace_far -name '*kernel_trace.csv' | head -1)
python $R/scripts/overlap_report.py "$T" | tee $R/$O/overlap_far.txt
python $R/scripts/frame_timeline.py "$T" | tee $R/$O/timeline_far.txt
grep '^{"metric"' $R/$O/trace_far.log | tail -1 | cut -c1-400
rm -rf $R/$O/trace_far
