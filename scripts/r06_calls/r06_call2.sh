#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_two_process.py -x -q -m gpu -s > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -5 $O/pytest_a.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-contract-ab $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
EXTRA=""
run default KT_X=1
run dense_level KT_DENSE_STEPWISE=0
run nogate KT_SIDE_GATE=0
EXTRA="--no-stress"
run orbit_gate1_step KT_SIDE_GATE=1 KT_ICP_LEVELS=0
run orbit_gate0_step KT_SIDE_GATE=0 KT_ICP_LEVELS=0
run orbit_gate1_level KT_SIDE_GATE=1
run orbit_default2 KT_X=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c2/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "odo_pipe", (j.get("stage_ms_pipelined") or {}).get("odometry"), "rc_pipe", (j.get("stage_ms_pipelined") or {}).get("raycast"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f gate %s" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0, s.get("side_gate")), "fallbacks", j["config"].get("odometry_fallbacks"))
PY
