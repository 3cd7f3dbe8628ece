# What-if variants of libkt_hip.so for bottleneck attribution (results of the variants are WRONG on purpose; never shipped).
# Build here (no GPU):   bash scripts/exp_variants.sh build "<flags for variant 1>" "<flags for variant 2>" ...
# Time on the GPU box:   bash scripts/exp_variants.sh run <workload> <steps>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $R/build/exp $R/exp
  i=0
  for flags in "$@"; do
    i=$((i+1))
    for f in kt_context kt_image kt_volume kt_track kt_tracker kt_hostmath; do
      src=$R/kintinuous_amd/csrc/$f.hip
      if [ $f = kt_volume ] || [ $f = kt_track ] || [ $f = kt_tracker ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c $src -o $R/build/exp/${f}_$i.o &
      fi
    done
    wait
    objs=""
    for f in kt_context kt_image kt_volume kt_track kt_tracker kt_hostmath kt_comm kt_slice kt_cloud kt_debug; do
      if [ -f $R/build/exp/${f}_$i.o ]; then objs="$objs $R/build/exp/${f}_$i.o"; else objs="$objs $R/build/$f.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/exp/libkt_exp_$i.so $objs
    echo "$i: $flags" 
  done > $R/exp/variants.txt
  cat $R/exp/variants.txt
else
  W=${1:-orbit512}; S=${2:-40}
  echo "== $W"
  python $R/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-readahead 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('base', d['roofline']['avg_launch_ms'], d['value'], d['stage_ms'])"
  while read line; do
    i=${line%%:*}
    KT_HIP_LIB=$R/exp/libkt_exp_$i.so python $R/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-readahead 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$line', d['roofline']['avg_launch_ms'], d['value'], d['stage_ms'])" || echo "$line FAILED"
  done < $R/exp/variants.txt
fi
