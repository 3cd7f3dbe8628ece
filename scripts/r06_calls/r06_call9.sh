#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c9; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_two_process.py tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_gpu_solve.py -x -q -m gpu -rs > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -6 $O/pytest_a.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-stress --no-contract-ab $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
for rep in a b; do
  EXTRA=""
  run one_$rep KT_X=1
  run three_$rep KT_ICP_ONE_LAUNCH=0
  (cd r05tree && timeout 600 python bench.py --no-cpu-baseline --no-contract-ab --no-stress > ../$O/bench_r05_$rep.json 2> ../$O/bench_r05_$rep.err; echo "r05 rc $?")
  EXTRA="--steps 20 --warmup 5"
  run drv_one_$rep KT_X=1
  (cd r05tree && timeout 600 python bench.py --no-cpu-baseline --no-contract-ab --no-stress --steps 20 --warmup 5 > ../$O/bench_drv_r05_$rep.json 2> ../$O/bench_drv_r05_$rep.err; echo "r05 rc $?")
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c9/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "odo_pipe", (j.get("stage_ms_pipelined") or {}).get("odometry"), "odo_serial", (j.get("stage_ms") or {}).get("odometry"), "p50", j["config"]["frame_ms"]["p50"])
PY
