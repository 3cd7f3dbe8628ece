#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c19; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_track.py tests/test_gpu_tracker.py tests/test_gpu_two_process.py tests/test_gpu_configs.py -x -q -m gpu -rs > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -6 $O/pytest_a.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-stress --no-contract-ab $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
for rep in a b; do
  EXTRA=""
  run fused_$rep KT_X=1
  run sep_$rep KT_ICP_FUSED_SETUP=0
  EXTRA="--steps 20 --warmup 5"
  run drv_fused_$rep KT_X=1
  run drv_sep_$rep KT_ICP_FUSED_SETUP=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c19/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "stages", j.get("stage_ms_pipelined"), "serial_odo", (j.get("stage_ms") or {}).get("odometry"), "p50", j["config"]["frame_ms"]["p50"], "plan", j["planned_frames"])
PY
