#!/bin/bash
# stream priorities: main high / side low, against all alike -- default bench (orbit512) and the dense view, A B A B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c21; mkdir -p $O
export TMPDIR=/tmp
python - <<'PY'
import ctypes
h=ctypes.CDLL("libamdhip64.so"); a=ctypes.c_int(); b=ctypes.c_int(); h.hipDeviceGetStreamPriorityRange(ctypes.byref(a),ctypes.byref(b)); print("priority range least",a.value,"greatest",b.value)
PY
for rep in 1 2; do for m in 0 1 2 3; do
  KT_STREAM_PRIORITY=$m timeout 900 python bench.py --no-cpu-baseline > $O/bench_p${m}_$rep.json 2> $O/bench_p${m}_$rep.err; echo "p$m rep$rep rc $?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c21/bench_p*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "stages", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
