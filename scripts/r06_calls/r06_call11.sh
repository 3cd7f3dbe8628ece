#!/bin/bash
# final-tree evidence: full suite, kernel statistics of the three workloads, PMC traffic of the voxel kernel on both roofline workloads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c11; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests/ -x -q -m gpu -rs > $O/pytest_full.log 2>&1 ) 2> $O/pytest_time.log; echo "pytest rc $?" >> $O/pytest_full.log
tail -4 $O/pytest_full.log
bash scripts/prof_bench.sh r06_final > $O/prof_bench.log 2>&1; tail -3 $O/prof_bench.log | cut -c1-200
bash scripts/prof_workload.sh farwall768 40 r06_farwall768 > $O/prof_far.log 2>&1
bash scripts/prof_workload.sh crabwalk512 200 r06_crabwalk512 > $O/prof_crab.log 2>&1
bash scripts/pmc_traffic.sh orbit512 16 > $O/traffic_orbit.log 2>&1; tail -1 $O/traffic_orbit.log | cut -c1-600
bash scripts/pmc_traffic.sh farwall768 6 > $O/traffic_far.log 2>&1; tail -1 $O/traffic_far.log | cut -c1-600
rm -rf gpurun_out/prof_r06_final gpurun_out/prof_r06_farwall768 gpurun_out/prof_r06_crabwalk512 gpurun_out/pmct_FETCH_SIZE gpurun_out/pmct_WRITE_SIZE
ls gpurun_out | head -40
