#!/bin/bash
# the driver's form of the bench (--steps 20 --warmup 5), five times, with the per-frame periods
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c30; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3 4 5; do
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress $EXTRA > $O/bench_d_$rep.json 2> $O/bench_d_$rep.err; echo "rep$rep rc $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c30/bench_d_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); c=j["config"]; r=j["roofline"]
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f n %s" % (r["frac"], r["frac_alone"] or 0, r.get("launches_timed")), c.get("frame_ms"), j.get("planned_frames"), j.get("host_ms_per_frame"), j.get("stage_ms_pipelined"))
PY
