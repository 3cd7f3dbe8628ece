#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c7; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-stress --no-contract-ab > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc $?"; }
for rep in a b; do
  run cur_$rep KT_X=1
  run sweep115_$rep KT_HIP_LIB=$R/variants/v_sweep115.so
  run noF_$rep KT_HIP_LIB=$R/variants/v_noF.so
  run noFnoclock_$rep KT_HIP_LIB=$R/variants/v_noF_noclock.so
  (cd r05tree && timeout 600 python bench.py --no-cpu-baseline --no-contract-ab --no-stress > ../$O/bench_r05_$rep.json 2> ../$O/bench_r05_$rep.err; echo "r05 rc $?")
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c7/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=j["roofline"]
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "odo_pipe", (j.get("stage_ms_pipelined") or {}).get("odometry"), "odo_serial", (j.get("stage_ms") or {}).get("odometry"))
PY
# r06 issue-roof counters: bit-exact and speed-of-light voxel kernels, both workloads
PASSES="sq sq2 tcc" bash scripts/pmc_issue.sh orbit512 16 1 r06_orbit_exact 2>&1 | grep PMCI | cut -c1-600
KT_TSDF_CONTRACT=sol PASSES="sq" bash scripts/pmc_issue.sh orbit512 16 1 r06_orbit_sol 2>&1 | grep PMCI | cut -c1-600
PASSES="sq sq2 tcc" bash scripts/pmc_issue.sh farwall768 6 1 r06_far_exact 2>&1 | grep PMCI | cut -c1-600
KT_TSDF_CONTRACT=sol PASSES="sq" bash scripts/pmc_issue.sh farwall768 6 1 r06_far_sol 2>&1 | grep PMCI | cut -c1-600
mkdir -p $R/$O/pmci; cp $R/gpurun_out/pmci_r06_*.json $R/$O/pmci/ 2>/dev/null
rm -rf $R/gpurun_out/pmci_r06_*/
