#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c32; mkdir -p $O
export TMPDIR=/tmp
for k in 20 30 24; do
  KT_BENCH_PERIODS=1 timeout 900 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-stress --no-contract-ab > $O/bench_k$k.json 2> $O/bench_k$k.err; echo "k$k rc $?"
  python - <<PY
import json
j=json.loads(open("$O/bench_k$k.json").read().strip().splitlines()[-1]); c=j["config"]
print("k$k fps %.0f" % j["value"], c.get("frame_ms"), j.get("planned_frames"))
print("   periods", c.get("periods_ms"))
PY
done
