#!/bin/bash
# the frame's plan made by the first call that knows the previous pose (the announce of the next frame): parity, then against the previous commit's library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c43; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_track.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_tol.py tests/test_gpu_host_shell.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for rep in 1 2 3; do
  timeout 900 python bench.py --no-cpu-baseline --no-contract-ab > $O/bench_early_$rep.json 2> $O/bench_early_$rep.err; echo "early $rep rc $?"
  KT_HIP_LIB=$GRAFT_REPO_ROOT/exp/libkt_prev.so timeout 900 python bench.py --no-cpu-baseline --no-contract-ab > $O/bench_prev_$rep.json 2> $O/bench_prev_$rep.err; echo "prev $rep rc $?"
done
for rep in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/drv_early_$rep.json 2> $O/drv_early_$rep.err
KT_HIP_LIB=$GRAFT_REPO_ROOT/exp/libkt_prev.so timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/drv_prev_$rep.json 2> $O/drv_prev_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c43/*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "plans", j.get("planned_frames"), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
