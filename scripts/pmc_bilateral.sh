# LDS / VALU counters of the bilateral kernels ("state its bound"): one --pmc pass, kernel-trace only, 20 launches each of
# kt_bilateral2_kernel and (KT_BILATERAL_V1=1) kt_bilateral_kernel at 640x480 -> gpurun_out/r05_pmc_bilateral.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/bil_run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from kintinuous_amd import abi, synth
ctx = abi.Ctx(0)
cam = synth.Camera()
_, frames, _, _ = synth.sequence("orbit", 2, cam)
d = np.ascontiguousarray(frames[1][0], np.uint16)
src, dst = ctx.upload(d), ctx.zeros(d.nbytes)
for _ in range(20):
    ctx.bilateral_filter(src, dst, cam.cols, cam.rows)
ctx.sync()
PY
: > $R/gpurun_out/r05_pmc_bilateral.txt
for v in 0 1; do
  rm -rf $R/gpurun_out/pmc_bil_$v
  KT_BILATERAL_V1=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bil_$v -- python /tmp/bil_run.py > $R/gpurun_out/pmc_bil_$v.log 2>&1 || tail -3 $R/gpurun_out/pmc_bil_$v.log
  python - <<PY >> $R/gpurun_out/r05_pmc_bilateral.txt
import csv, glob, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob("$R/gpurun_out/pmc_bil_$v/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "bilateral" in r["Kernel_Name"] and "lut" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$R/gpurun_out/pmc_bil_$v/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "bilateral" in r["Kernel_Name"] and "lut" not in r["Kernel_Name"]:
            dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print("KT_BILATERAL_V1=$v", {k: round(sum(x) / len(x)) for k, x in acc.items()}, "launches", len(dur), "mean us under the counters", round(sum(dur) / max(1, len(dur)), 1))
PY
done
cat $R/gpurun_out/r05_pmc_bilateral.txt
