# r04 call 7: slice stage with larger cell radii (parity + timing), nt cache policy of the volume accesses on the dense case (time + PMC)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### parity: slice / pcd / host shell"
timeout 900 python -m pytest tests/test_slice_process.py tests/test_pcd.py tests/test_gpu_host_shell.py -m gpu -q > gpurun_out/c7_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c7_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/c7_tests.log | head -20
echo "#### slice stage"
python scripts/slice_stage_timing.py 2>/dev/null | tail -4
echo "#### nt variants: time"
L=$PWD/kintinuous_amd
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric')][-1]); r=d['roofline']; print('$1', round(d['value'],1), 'tsdf23 %.4f ms, alone %.4f, frac %.4f' % (r['avg_launch_ms'], r['avg_launch_ms_alone'], r['frac']))"; }
for rep in 1 2; do
  for i in 0 1 2 3; do
    lib=""; [ $i != 0 ] && lib=$L/libkt_exp_$i.so
    KT_HIP_LIB=$lib python bench.py --workload farwall768 --steps 12 --warmup 3 --no-cpu-baseline --no-readahead 2>/dev/null | line "far v$i"
  done
done
for i in 0 1 2; do
  lib=""; [ $i != 0 ] && lib=$L/libkt_exp_$i.so
  KT_HIP_LIB=$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress 2>/dev/null | line "orbit v$i"
done
echo "#### nt variants: traffic"
bash scripts/pmc_variants.sh 2>&1 | grep PMC | grep farwall
