#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c17; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_image.py tests/test_golden_ref.py tests/test_golden.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -3 $O/pytest_a.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-stress --no-contract-ab $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
for rep in a b; do
  EXTRA=""
  run s2_$rep KT_X=1
  run s4_$rep KT_PYR_S=4
  EXTRA="--steps 20 --warmup 5"
  run drv_s2_$rep KT_X=1
  run drv_s4_$rep KT_PYR_S=4
done
EXTRA="--workload farwall768 --steps 40 --warmup 10"
run far_s2 KT_X=1
run far_s4 KT_PYR_S=4
EXTRA="--workload crabwalk512"
run crab_s2 KT_X=1
run crab_s4 KT_PYR_S=4
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c17/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "stages", j.get("stage_ms_pipelined"), "pyr_serial", (j.get("stage_ms") or {}).get("pyramid"), "p50", j["config"]["frame_ms"]["p50"])
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_orbit -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-contract-ab --no-stress > $R/$O/trace_orbit.log 2>&1
T=$(find $R/$O/trace_orbit -name '*kernel_trace.csv' | head -1)
python $R/scripts/stream_timeline.py "$T" | tee $R/$O/stream_timeline_orbit.txt
rm -rf $R/$O/trace_orbit
