# A/B of the libkt_exp_<i>.so variants built by `scripts/exp_variants.sh build ...` (exp/variants.txt) against the tree's
# library, in one GPU call: parity verdict (volume / sweep / golden tests through KT_HIP_LIB) and tsdf23 launch times on both workloads.
#   gpurun --timeout 900 -- 'bash scripts/variants_ab.sh [steps]'
cd "${GRAFT_REPO_ROOT:-.}"
S=${1:-40}
one() {   # $1 = label, $2 = library path or empty
  echo "== $1"
  KT_HIP_LIB=$2 python -m pytest tests/test_gpu_volume.py tests/test_gpu_sweep.py tests/test_golden_ref.py tests/test_golden.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -1
  for w in orbit512 farwall768; do
    KT_HIP_LIB=$2 python bench.py --workload $w --steps $S --warmup 10 --no-cpu-baseline --no-stress --no-readahead 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   $w: %.0f fps, tsdf23 %.1f us in the frame, %.1f us alone, frac_alone %.3f, lane_eff %s, stage %s' % (d['value'], 1e3*r['avg_launch_ms'], 1e3*r['avg_launch_ms_alone'], r['frac_alone'], r.get('lane_efficiency'), d['stage_ms']))" || echo "   $w FAILED"
  done
}
[ -z "$KT_SKIP_TREE" ] && one "tree" ""
while read line; do
  i=${line%%:*}
  case "$line" in *WHATIF*) continue;; esac
  one "variant $line" $PWD/exp/libkt_exp_$i.so
done < exp/variants.txt
