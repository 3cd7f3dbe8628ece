#!/bin/bash
# the plan's completion as a word the set-up kernel waits for (no event wait in front of it): parity, then the bench against the previous commit's library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c42; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_track.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_tol.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for rep in 1 2 3; do
  timeout 900 python bench.py --no-cpu-baseline --no-contract-ab > $O/bench_word_$rep.json 2> $O/bench_word_$rep.err; echo "word $rep rc $?"
  KT_HIP_LIB=$GRAFT_REPO_ROOT/exp/libkt_prev.so timeout 900 python bench.py --no-cpu-baseline --no-contract-ab > $O/bench_event_$rep.json 2> $O/bench_event_$rep.err; echo "event $rep rc $?"
done
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/drv_word.json 2> $O/drv_word.err
KT_HIP_LIB=$GRAFT_REPO_ROOT/exp/libkt_prev.so timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/drv_event.json 2> $O/drv_event.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c42/*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "plans", j.get("planned_frames"), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
