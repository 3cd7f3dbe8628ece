#!/bin/bash
# final tree: the headline ten times in each form (spread of the numbers quoted), and bigger soaks (320x240 into 160^3 and 192^3, every mode)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c46; mkdir -p $O
export TMPDIR=/tmp
: > $O/r06_bench_repeats.jsonl
for rep in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python bench.py --no-cpu-baseline --no-stress --no-contract-ab 2> $O/rep_def_$rep.err | tail -1 >> $O/r06_bench_repeats.jsonl
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-contract-ab 2> $O/rep_drv_$rep.err | tail -1 >> $O/r06_bench_repeats.jsonl
done
python - <<'PY'
import json, statistics as st
rows=[json.loads(l) for l in open("gpurun_out/c46/r06_bench_repeats.jsonl") if l.startswith("{")]
for steps in (200, 20):
    r=[x for x in rows if x["steps"]==steps]
    v=[x["value"] for x in r]; f=[x["roofline"]["frac"] for x in r]; fa=[x["roofline"]["frac_alone"] for x in r]
    print("steps %d: n %d fps mean %.0f sd %.0f min %.0f max %.0f | frac mean %.3f min %.3f max %.3f | alone mean %.3f" % (steps, len(v), st.mean(v), st.pstdev(v), min(v), max(v), st.mean(f), min(f), max(f), st.mean(fa)))
PY
for mode in icp rgbd_icp rgbd; do
  timeout 1500 python tests/tools/soak.py $mode 320 160 2>&1 | tail -2 | tee -a $O/soak_big.log
done
timeout 1500 python tests/tools/soak.py icp 320 192 2>&1 | tail -2 | tee -a $O/soak_big.log
