#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c33; mkdir -p $O
export TMPDIR=/tmp
KT_PLAN_TRACE=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-contract-ab > $O/trace.json 2> $O/trace.err; grep "^plan" $O/trace.err | head -32
for fs in 1 2 3; do for rep in 1 2; do
  KT_PLAN_FLOOR_SCALE=$fs timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-contract-ab > $O/drv_fs${fs}_$rep.json 2> $O/drv_fs${fs}_$rep.err
done; KT_PLAN_FLOOR_SCALE=$fs timeout 900 python bench.py --no-cpu-baseline --no-contract-ab > $O/def_fs${fs}.json 2> $O/def_fs${fs}.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c33/d*_fs*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); c=j["config"]; r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), c["frame_ms"]["slowest"][:2], j.get("planned_frames"), "tsdf23 alone ms", r.get("avg_launch_ms_alone"), "| stress pipe %.3f frame %.3f" % (s.get("frac_pipelined") or 0, s.get("frame_ms_pipelined") or 0))
PY
