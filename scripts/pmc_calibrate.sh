cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cal_$c -- python $R/scripts/pmc_calibrate.py > $R/gpurun_out/pmc_cal_$c.log 2>&1
done
python - <<PY
import csv,glob,collections
for c in ['FETCH_SIZE','WRITE_SIZE']:
    f=glob.glob('$R/gpurun_out/pmc_cal_%s/*/*counter_collection.csv'%c)[0]
    rows=[r for r in csv.DictReader(open(f)) if 'kt_stream_kernel' in r['Kernel_Name']]
    for r in rows: print(c, r['Kernel_Name'][:45], r['Counter_Value'])
PY
