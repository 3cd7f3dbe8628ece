#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_two_process.py "tests/test_gpu_tracker.py::test_icp_chain_per_level_equals_per_iteration" "tests/test_gpu_tracker.py::test_the_level_form_needs_to_be_alone" tests/test_gpu_tol.py tests/test_gpu_configs.py -x -q -m gpu -s > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -5 $O/pytest_a.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
EXTRA=""
run default KT_X=1
EXTRA="--no-stress --no-contract-ab"
for rep in a b; do
  run cur_$rep KT_X=1
  (cd r05tree && timeout 600 python bench.py --no-cpu-baseline --no-contract-ab --no-stress > ../$O/bench_r05_$rep.json 2> ../$O/bench_r05_$rep.err; echo "r05 rc $?")
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c5/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "odo_pipe", (j.get("stage_ms_pipelined") or {}).get("odometry"), "odo_serial", (j.get("stage_ms") or {}).get("odometry"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f gate %s sol %s" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0, s.get("side_gate"), s.get("speed_of_light")), "ab", r.get("contract_ab"))
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_far -- python $R/bench.py --workload farwall768 --steps 12 --warmup 8 --no-cpu-baseline --no-contract-ab > $R/$O/trace_far.log 2>&1
T=$(find $R/$O/trace_far -name '*kernel_trace.csv' | head -1)
python $R/scripts/overlap_report.py "$T" | tee $R/$O/overlap_far.txt
python $R/scripts/frame_timeline.py "$T" | tee $R/$O/timeline_far.txt
rm -rf $R/$O/trace_far
