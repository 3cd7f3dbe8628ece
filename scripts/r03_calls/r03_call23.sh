cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### volume (extract) / tracker / configs / host shell / pcd / slice tests"
python -m pytest tests/test_gpu_volume.py tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_gpu_host_shell.py tests/test_pcd.py tests/test_slice_process.py -m gpu -q > gpurun_out/call23_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/call23_tests.log | tail -3
echo "#### smoke, then driver-style runs (the driver's order)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for rep in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_driverstyle_$rep.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_driverstyle_$rep.json')); print(round(d['value'],1), d['ms_per_step'], 'frac', round(d['roofline']['frac'],4), 'ratio', d['roofline'].get('traffic_ratio'), d['config']['frame_ms'])"; done
