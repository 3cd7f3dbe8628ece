cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$PWD/kintinuous_amd
echo "#### tracker / configs tests (set-up kernel split)"
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/call12_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call12_tests.log | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('roofline_stress') or {}; print('$1', round(d['value'],1), 'pipe', d.get('stage_ms_pipelined'), 'tsdf23 %.1f us alone %.1f frac %.4f lane %s stress %.4f ms' % (1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac'], r.get('lane_efficiency'), s.get('avg_launch_ms', 0)))"; }
echo "#### set-up split A/B (2 = HEAD)"
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit tree"
  KT_HIP_LIB=$L/libkt_exp_2.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit head"
done
echo "#### 16x4 wave-columns (variant 1): parity, then both workloads"
KT_HIP_LIB=$L/libkt_exp_1.so python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -1
KT_HIP_LIB=$L/libkt_exp_1.so python -m pytest tests/test_gpu_tracker.py -m gpu -q -k "planned or readahead or shift" 2>&1 | grep -E "passed|failed|error" | tail -1
for rep in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "32x2"
  KT_HIP_LIB=$L/libkt_exp_1.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line "16x4"
done
echo "#### driver-style"
for rep in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stress 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], round(d['roofline']['frac'],4), d['roofline']['launches_timed'], d['planned_frames'], d['config']['frame_ms'])"; done
