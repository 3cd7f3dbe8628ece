# memory-path counters for the tsdf23 / raycast kernels (one pass per group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_$n -- python $R/bench.py --workload orbit512 --steps 30 --warmup 2 --no-cpu-baseline --no-readahead --no-stress > $R/gpurun_out/pmcm_$n.log 2>&1 || tail -3 $R/gpurun_out/pmcm_$n.log
}
run a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
run b TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
python - <<PY
import csv,glob,collections,os
for n in ['a','b']:
    fs=sorted(glob.glob('$R/gpurun_out/pmcm_%s/*/*counter_collection.csv'%n), key=os.path.getmtime)
    if not fs: print(n,'no output'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k=r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        if 'tsdf23_kernel<false' in k or 'raycast_kernel<false' in k:
            print(n, k, {c: round(sum(v)/len(v),1) for c,v in acc[k].items()})
PY
