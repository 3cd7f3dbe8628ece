#!/bin/bash
# final tree (after the read-ahead stage work): PMC traffic of the voxel kernel for the new kt_volume.hip, then the bench lines and the workloads table
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c40; mkdir -p $O
export TMPDIR=/tmp
bash scripts/pmc_traffic.sh orbit512 16 2>&1 | tail -1 | cut -c1-400
bash scripts/pmc_traffic.sh farwall768 6 2>&1 | tail -1 | cut -c1-400
rm -rf gpurun_out/pmct_FETCH_SIZE gpurun_out/pmct_WRITE_SIZE
cp gpurun_out/r06_pmc_tsdf23_orbit512.json gpurun_out/r06_pmc_tsdf23_farwall768.json profiles/ 2>/dev/null   # (so that the bench lines below quote them)
cp gpurun_out/r06_pmc_tsdf23_*.json $O/
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err; echo "default rc $?"
timeout 900 python bench.py --no-cpu-baseline > $O/r06_bench_default_2.json 2> $O/bench_default2.err; echo "default2 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driverstyle_1.json 2> $O/bench_drv1.err; echo "drv1 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_driverstyle_2.json 2> $O/bench_drv2.err; echo "drv2 rc $?"
timeout 900 python bench.py --host-frames --no-cpu-baseline --no-stress > $O/r06_bench_hostframes.json 2> $O/bench_host.err; echo "host rc $?"
bash scripts/run_all_workloads.sh > $O/workloads.log 2>&1; cp gpurun_out/workloads_r06.jsonl $O/r06_workloads.jsonl; tail -5 $O/workloads.log | cut -c1-200
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c40/r06_bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "lane", r.get("lane_efficiency"), "traffic_ratio", r.get("traffic_ratio"), "serial", j.get("stage_ms"), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / %.3f sol %s tr %s" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0, (s.get("speed_of_light") or {}).get("frac_alone"), s.get("traffic_ratio")), "cpu", (j.get("cpu_baseline") or {}).get("value"))
PY
