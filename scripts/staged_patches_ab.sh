# A/B of the patches staged under scripts/ against the tree, in ONE gpurun call (run from the repo root on the GPU box):
#   gpurun --timeout 420 -- 'bash scripts/staged_patches_ab.sh tsdf_pk.patch weight_clamp.patch'
# Prints, for the tree and for the tree + patches: the parity verdict of the volume / sweep / golden tests and the tsdf23 launch time
# (HIP events, undisturbed) on orbit512 and farwall768.  Leaves the working tree as it found it.  Ship a patch only if both lines of its
# run say "passed" and the time is lower; then re-collect the PMC traffic (scripts/pmc_traffic.sh) -- bench.py quotes it by file hash.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  echo "== $1"
  python -m kintinuous_amd.build > /dev/null 2>&1 || { echo "build failed"; return; }
  python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -1
  for w in orbit512 farwall768; do
    python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-stress --no-readahead 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   $w: %.0f fps, tsdf23 %.1f us in the frame, %.1f us alone, frac_alone %.3f' % (d['value'], 1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac_alone']))"
  done
}
run "tree as committed"
for p in "$@"; do git apply "scripts/$p" || echo "cannot apply $p"; done
run "tree + $*"
for p in "$@"; do git apply -R "scripts/$p" 2>/dev/null; done
python -m kintinuous_amd.build > /dev/null 2>&1
