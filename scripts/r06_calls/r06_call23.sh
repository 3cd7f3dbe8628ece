#!/bin/bash
# read-ahead stage: scale_depth from an LDS window, pyramid in two launches -- parity, then kernel times under rocprofv3, then the bench (forms A/B)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c23; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_image.py tests/test_gpu_volume.py tests/test_gpu_configs.py tests/test_gpu_sweep.py tests/test_gpu_tracker.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
KT_PYR_FORM=3 timeout 600 python -m pytest tests/test_gpu_image.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_f3.log 2>&1; echo "pytest form3 rc $?"; tail -2 $O/pytest_f3.log
for f in 1 2 3; do
  ( cd /tmp && KT_PYR_FORM=$f rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_f$f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-stress > $GRAFT_REPO_ROOT/$O/prof_f$f.json 2> $GRAFT_REPO_ROOT/$O/prof_f$f.err )
  python - <<PY
import csv,glob
fs=glob.glob("$O/prof_f$f/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(fs[0])))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("pyramid","scale_depth","tile_finish","bilateral2")): print("form $f", n[:50].ljust(50), "calls", r["Calls"], "avg %.1f min %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
for rep in 1 2; do for f in 1 2 3; do
  KT_PYR_FORM=$f timeout 900 python bench.py --no-cpu-baseline > $O/bench_f${f}_$rep.json 2> $O/bench_f${f}_$rep.err; echo "f$f rep$rep rc $?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c23/bench_f*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "serial", j.get("stage_ms"), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0), (s.get("stage_ms") or ""))
PY
