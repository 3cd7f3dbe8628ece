#!/bin/bash
# posted-word gate (KT_SIDE_GATE=3: the ray cast posts a word, the read-ahead stream's command processor waits on it) against the default, A B A B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c22; mkdir -p $O
export TMPDIR=/tmp
KT_SIDE_GATE=3 timeout 600 python -m pytest tests/test_gpu_tracker.py -x -q -m gpu > $O/pytest_g3.log 2>&1; echo "pytest gate3 rc $?"; tail -3 $O/pytest_g3.log
for rep in 1 2; do for m in 2 3; do
  KT_SIDE_GATE=$m timeout 900 python bench.py --no-cpu-baseline > $O/bench_g${m}_$rep.json 2> $O/bench_g${m}_$rep.err; echo "g$m rep$rep rc $?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c22/bench_g*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "stages", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
