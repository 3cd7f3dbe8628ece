#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c20; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu -rs > $O/pytest_full.log 2>&1 ) 2> $O/pytest_time.log; echo "pytest rc $?" >> $O/pytest_full.log
tail -4 $O/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/c20/bench_default.json","gpurun_out/c20/bench_driver.json"):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "traffic", r.get("traffic_ratio"), "stages", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
