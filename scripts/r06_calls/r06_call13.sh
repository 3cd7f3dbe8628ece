#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c13; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_track.py -x -q -m gpu -rs > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" >> $O/pytest_a.log
tail -6 $O/pytest_a.log
timeout 600 python bench.py --workload orbit256 --no-cpu-baseline --no-stress --no-contract-ab > $O/bench_orbit256.json 2> $O/bench_orbit256.err; echo "orbit256 rc $?"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_orbit -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-contract-ab --no-stress > $R/$O/trace_orbit.log 2>&1
T=$(find $R/$O/trace_orbit -name '*kernel_trace.csv' | head -1)
python $R/scripts/stream_timeline.py "$T" | tee $R/$O/stream_timeline_orbit.txt
python $R/scripts/overlap_report.py "$T" "kt_tsdf23_lean_kernel<false" | tee $R/$O/overlap_orbit_tsdf.txt
python $R/scripts/overlap_report.py "$T" "kt_raycast_kernel<false" | tee $R/$O/overlap_orbit_rc.txt
python $R/scripts/overlap_report.py "$T" "kt_icp_level_kernel" | tee $R/$O/overlap_orbit_icp.txt
rm -rf $R/$O/trace_orbit
cd $R
python -c "
import json
j=json.loads(open('gpurun_out/c13/bench_orbit256.json').read().strip().splitlines()[-1]); print('orbit256 fps', round(j['value']), j['config'].get('side_gate'), j['stage_ms_pipelined'])"
