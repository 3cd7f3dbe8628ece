# Round 4: the counters behind profiles/r04_issue_roof.md -- for the voxel kernel (either variant) on one workload:
#   pass sq   : SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
#   pass sq2  : SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
#   pass grbm : GRBM_GUI_ACTIVE (cycles the chip was busy: / the kernel's duration = the shader clock under this kernel)
#   pass tcc  : TCC_HIT_sum TCC_MISS_sum   (L2 hit rate)
# PASSES="sq tcc" in the environment selects a subset.
# kernel-trace only, one pass per group.   usage: pmc_issue.sh <workload> <steps> <lean 0|1> <tag>
cd /tmp && export TMPDIR=/tmp
W=${1:-orbit512}; S=${2:-16}; L=${3:-0}; T=${4:-x}
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  n=$1; shift
  rm -rf $R/gpurun_out/pmci_${T}_$n
  KT_TSDF_LEAN=$L rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmci_${T}_$n -- python $R/bench.py --workload $W --steps $S --warmup 2 --no-cpu-baseline --no-contract-ab --no-readahead --no-stress > $R/gpurun_out/pmci_${T}_$n.log 2>&1 || { echo "pass $n FAILED"; tail -3 $R/gpurun_out/pmci_${T}_$n.log; }
}
P=${PASSES:-sq sq2 grbm tcc}
want() { case " $P " in *" $1 "*) return 0;; esac; return 1; }
want sq && run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
want sq2 && run sq2 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
want grbm && run grbm GRBM_GUI_ACTIVE
want tcc && run tcc TCC_HIT_sum TCC_MISS_sum
python - <<PY
import csv, glob, collections, json
out = {}
for n in ['sq', 'sq2', 'grbm', 'tcc']:
    fs = glob.glob('$R/gpurun_out/pmci_${T}_%s/*/*counter_collection.csv' % n)
    ks = glob.glob('$R/gpurun_out/pmci_${T}_%s/*/*kernel_trace.csv' % n)
    if not fs:
        print(n, 'no file'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0][-44:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    if ks:
        for r in csv.DictReader(open(ks[0])):
            dur[r['Kernel_Name'].split('(')[0][-44:]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    for k in acc:
        if ('tsdf23' in k and '<false' in k) or 'raycast_kernel<false' in k:
            row = {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}
            row['launches'] = len(next(iter(acc[k].values())))
            if dur.get(k): row['mean_duration_ns_this_pass'] = round(sum(dur[k]) / len(dur[k]), 1)
            out.setdefault(k, {}).update({n + ':' + c: v for c, v in row.items()})
            print('PMCI', '$T', '$W', 'lean=$L', n, k, row)
json.dump(out, open('$R/gpurun_out/pmci_${T}.json', 'w'), indent=1)
PY
