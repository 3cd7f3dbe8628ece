# PMC passes for the tsdf23 kernel (separate passes, kernel-trace only): usage pmc_tsdf.sh <workload> <steps>
cd /tmp && export TMPDIR=/tmp
W=${1:-farwall768}; S=${2:-6}
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -- python $R/bench.py --workload $W --steps $S --warmup 2 --no-cpu-baseline --no-readahead --no-stress > $R/gpurun_out/pmc_$n.log 2>&1
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<PY
import csv,glob,collections
for n in ['sq','sq2','fetch','write']:
    fs=glob.glob('$R/gpurun_out/pmc_%s/*/*counter_collection.csv'%n)
    if not fs: print(n,'no file'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        if 'tsdf23' in k or 'raycast' in k or 'icp' in k:
            print(n, k, {c: round(sum(v)/len(v),1) for c,v in acc[k].items()}, 'n=',len(next(iter(acc[k].values()))))
PY
