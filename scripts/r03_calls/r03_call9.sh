cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### rgbd tests"; python -m pytest tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_gpu_track.py -m gpu -q -k "rgbd or rgb or config3 or joint or readahead or planned" 2>&1 | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'odo', d['stage_ms']['odometry'], 'pipe', d.get('stage_ms_pipelined'), d['config']['frame_ms'])"; }
for rep in 1 2; do
  python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab tree"
  KT_HIP_LIB=$PWD/kintinuous_amd/libkt_exp_1.so python bench.py --workload crabwalk512 --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "crab old"
done
echo "#### driver-style"; for rep in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], round(d['roofline']['frac'],4), d['roofline']['launches_timed'], d['planned_frames'])"; done
echo "#### pmc traffic orbit"; bash scripts/pmc_traffic.sh orbit512 16 2>&1 | tail -1 | cut -c1-900
