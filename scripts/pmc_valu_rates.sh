# Round 4: the SQ counters of the issue-rate micro-benchmark itself (scripts/valu_rates.py), so that SQ_ACTIVE_INST_* / SQ_INSTS_* of a
# KNOWN instruction stream calibrate what those counters mean for the voxel kernel.   -> gpurun_out/pmc_valu_rates.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -c1-8)
  rm -rf $R/gpurun_out/pmcvr_$tag
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcvr_$tag -- python $R/scripts/valu_rates.py > $R/gpurun_out/pmcvr_$tag.log 2>&1 || tail -3 $R/gpurun_out/pmcvr_$tag.log
done
python - <<PY
import csv, glob, collections
rows = collections.OrderedDict()
for tag in ['SQ_WAVE_', 'GRBM_GUI']:
    fs = glob.glob('$R/gpurun_out/pmcvr_%s/*/*counter_collection.csv' % tag)
    ks = glob.glob('$R/gpurun_out/pmcvr_%s/*/*kernel_trace.csv' % tag)
    if not fs: print(tag, 'no file'); continue
    dur = {}
    for r in csv.DictReader(open(ks[0])):
        dur[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp']), r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'))
    for r in csv.DictReader(open(fs[0])):
        if 'valu_rate' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'].split('(')[0][-28:], r['Dispatch_Id'])
        rows.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
        if r['Dispatch_Id'] in dur: rows[key]['ns_' + tag] = dur[r['Dispatch_Id']][0]; rows[key]['grid'] = dur[r['Dispatch_Id']][1]
with open('$R/gpurun_out/pmc_valu_rates.txt', 'w') as f:
    for k, v in rows.items():
        line = '%s disp %s %s' % (k[0], k[1], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
        f.write(line + '\n')
print(open('$R/gpurun_out/pmc_valu_rates.txt').read()[:6000])
PY
