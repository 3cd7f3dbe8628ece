#!/bin/bash
# plan of the CURRENT frame (one increment) against the read-ahead frame's (two increments): parity, then bench A B A B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c28; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_track.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for rep in 1 2; do for m in 0 1; do
  KT_PLAN_CURRENT=$m timeout 900 python bench.py --no-cpu-baseline > $O/bench_pc${m}_$rep.json 2> $O/bench_pc${m}_$rep.err; echo "pc$m rep$rep rc $?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c28/bench_pc*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "plans", j.get("planned_frames"), "U", r.get("units_per_launch") or r.get("U"), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0), s.get("planned_frames"))
PY
