#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c10; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest "tests/test_gpu_tracker.py::test_ri_chain_per_level_equals_per_iteration" "tests/test_gpu_tracker.py::test_the_level_form_needs_to_be_alone" "tests/test_gpu_tracker.py::test_icp_chain_per_level_equals_per_iteration" -x -q -m gpu -rs > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -15 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -rs -k "config3 or crabwalk" > $O/pytest_b.log 2>&1; echo "pytest_b rc $?" >> $O/pytest_b.log
tail -5 $O/pytest_b.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --workload crabwalk512 --no-cpu-baseline --no-stress --no-contract-ab > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
for rep in a b; do
  run ri_level_$rep KT_X=1
  run ri_step_$rep KT_RI_LEVELS=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c10/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "fps %.0f" % j["value"], "stages", j.get("stage_ms_pipelined"), "serial", j.get("stage_ms"), "fallbacks", j["config"].get("odometry_fallbacks"), "p50", j["config"]["frame_ms"]["p50"])
PY
