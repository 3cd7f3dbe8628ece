#!/bin/bash
# the voxel kernel's fixed grid: 8192 waves (every slot) against 7168 / 6144 (slack for the side streams' workgroups), A B C A B C
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c27; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 900 python bench.py --no-cpu-baseline > $O/bench_w8192_$rep.json 2> $O/bench_w8192_$rep.err; echo "8192 rep$rep rc $?"
  for w in 7168 6144; do
    KT_HIP_LIB=$R/exp/libkt_w$w.so timeout 900 python bench.py --no-cpu-baseline > $O/bench_w${w}_$rep.json 2> $O/bench_w${w}_$rep.err; echo "$w rep$rep rc $?"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c27/bench_w*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
