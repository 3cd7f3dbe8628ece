# HBM-side traffic of kt_tsdf23_kernel for the tree's library and every variant of exp/variants.txt: FETCH_SIZE (all) and
# WRITE_SIZE (tree + non-what-if variants), one --pmc pass each, kernel-trace only.  Prints KiB per launch (mean) per {lib, workload}.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # label lib counter workload steps
  rm -rf $R/gpurun_out/pmcv
  KT_HIP_LIB=$2 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcv -- python $R/bench.py --workload $4 --steps $5 --warmup 2 --no-cpu-baseline --no-readahead --no-stress > $R/gpurun_out/pmcv.log 2>&1 || tail -3 $R/gpurun_out/pmcv.log
  python - <<PY
import csv, glob
fs = glob.glob("$R/gpurun_out/pmcv/*/*counter_collection.csv")
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "tsdf23" in r["Kernel_Name"] and "<false" in r["Kernel_Name"] and r["Counter_Name"] == "$3"] if fs else []
print("PMC %-40s %-11s %-10s launches %3d  mean %.1f KiB  (x1024: %.2f MB)" % ("$1", "$4", "$3", len(vals), sum(vals) / max(1, len(vals)), sum(vals) / max(1, len(vals)) * 1024 / 1e6))
PY
}
for w in orbit512 farwall768; do
  S=16; [ $w = farwall768 ] && S=6
  run tree "" FETCH_SIZE $w $S
  run tree "" WRITE_SIZE $w $S
  while read line; do
    i=${line%%:*}
    run "variant $line" $R/exp/libkt_exp_$i.so FETCH_SIZE $w $S
    case "$line" in *WHATIF*) ;; *) run "variant $line" $R/exp/libkt_exp_$i.so WRITE_SIZE $w $S;; esac
  done < $R/exp/variants.txt
done
