#!/bin/bash
# One independent RGB-D stream per GPU (replicas only, no data-path collective), N processes on one node:
#   bench:   scripts/launch_8gpu.sh bench [N] [steps] [warmup]         -> bench.py under torch.distributed.run, one JSON line from rank 0
#   logs:    scripts/launch_8gpu.sh logs  [N] log0.klg log1.klg ...    -> kintinuous_hip per GPU, poses gathered at the end (kt_pose_gather)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0
mode=${1:-bench}; N=${2:-8}
if [ "$mode" = bench ]; then
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${PORT:-29533} $R/bench.py --gpus $N --steps ${3:-200} --warmup ${4:-20}
fi
shift 2
comm=$(mktemp -u /tmp/kt_comm_XXXXXX)
pids=()
for r in $(seq 0 $((N - 1))); do
  log=${1:?one log per rank}; shift
  $R/kintinuous_amd/host/bin/kintinuous_hip -l $log -g $r -rank $r -world $N -comm $comm -o /tmp/kt_stream_$r &
  pids+=($!)
done
# the ranks meet in ONE collective at the end: if any of them fails first, the others would wait for it inside RCCL -- end them
rc=0
left=${#pids[@]}
while [ $left -gt 0 ]; do
  if wait -n; then :; else rc=1; kill "${pids[@]}" 2>/dev/null || true; fi
  left=$((left - 1))
done
rm -f $comm
exit $rc
