#!/bin/bash
# read-ahead protocols on ONE box, interleaved: first (announce in front, waits for the pose: rounds 2-5), behind1, behind2; sparse + dense legs, host frames
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c39; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_host_shell.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
run() { name=$1; shift; env "$@" timeout 900 python bench.py --no-cpu-baseline --no-contract-ab > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
for rep in 1 2 3; do
  run first_$rep KT_BENCH_ANNOUNCE_FIRST=1 KT_PREFETCH_WAIT=1
  run behind1_$rep KT_BENCH_AHEAD=1
  run behind2_$rep KT_BENCH_AHEAD=2
done
KT_BENCH_ANNOUNCE_FIRST=1 KT_PREFETCH_WAIT=1 timeout 900 python bench.py --host-frames --no-cpu-baseline --no-stress > $O/host_first.json 2> $O/host_first.err
KT_BENCH_AHEAD=1 timeout 900 python bench.py --host-frames --no-cpu-baseline --no-stress > $O/host_behind1.json 2> $O/host_behind1.err
KT_BENCH_AHEAD=2 timeout 900 python bench.py --host-frames --no-cpu-baseline --no-stress > $O/host_behind2.json 2> $O/host_behind2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c39/*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
