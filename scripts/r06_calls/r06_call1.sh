#!/bin/bash
# round 6, GPU call 1: the safety changes of the ICP level kernel + the side-stream gate A/B + cooperative-launch A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_two_process.py "tests/test_gpu_tracker.py::test_icp_chain_per_level_equals_per_iteration" "tests/test_gpu_tracker.py::test_the_level_form_needs_to_be_alone" -x -q -m gpu -s > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -5 $O/pytest_a.log
for g in 2 0 1; do
  KT_SIDE_GATE=$g timeout 600 python bench.py --no-cpu-baseline > $O/bench_gate$g.json 2> $O/bench_gate$g.err; echo "gate $g rc $?"
done
KT_ICP_COOP=1 timeout 300 python bench.py --no-cpu-baseline --no-stress --no-contract-ab > $O/bench_coop.json 2> $O/bench_coop.err; echo "coop rc $?"
timeout 300 python bench.py --no-cpu-baseline --no-stress --no-contract-ab --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c1/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "odo_pipe", (j.get("stage_ms_pipelined") or {}).get("odometry"), "rc_pipe", (j.get("stage_ms_pipelined") or {}).get("raycast"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f gate %s" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0, s.get("side_gate")), "fallbacks", j["config"].get("odometry_fallbacks"))
PY
