# FETCH_SIZE / WRITE_SIZE against known byte counts (scripts/pmc_calibrate.py): one --pmc pass per counter, kernel-trace only.
# Prints one line per launch {counter, kernel, KiB counted}; the caller copies the csv files into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cal_$c -- python $R/scripts/pmc_calibrate.py > $R/gpurun_out/pmc_cal_$c.log 2>&1
done
python - <<PY
import csv,glob,os
for c in ['FETCH_SIZE','WRITE_SIZE']:
    f=sorted(glob.glob('$R/gpurun_out/pmc_cal_%s/*/*counter_collection.csv'%c), key=os.path.getmtime)[-1]
    rows=[r for r in csv.DictReader(open(f)) if 'kt_stream' in r['Kernel_Name']]
    for r in rows: print(c, r['Kernel_Name'][:60], r['Counter_Value'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us')
    os.system('cp %s $R/gpurun_out/r03_pmc_calibration_%s.csv' % (f, c.lower().split('_')[0]))
PY
