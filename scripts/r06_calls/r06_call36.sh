#!/bin/bash
# announce-behind everywhere (bench host-frames path, dense-view leg, the host shell): parity, then A B A B incl. --host-frames
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c36; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_host_shell.py tests/test_gpu_configs.py tests/test_gpu_track.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for rep in 1 2; do for m in 1 0; do
  if [ $m = 1 ]; then export KT_BENCH_ANNOUNCE_FIRST=1; else unset KT_BENCH_ANNOUNCE_FIRST; fi
  timeout 900 python bench.py --no-cpu-baseline > $O/bench_af${m}_$rep.json 2> $O/bench_af${m}_$rep.err; echo "af$m rep$rep rc $?"
  timeout 900 python bench.py --host-frames --no-cpu-baseline --no-stress > $O/host_af${m}_$rep.json 2> $O/host_af${m}_$rep.err
done; done
unset KT_BENCH_ANNOUNCE_FIRST
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c36/*_af*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0))
PY
