# HBM-side traffic of the tsdf23 kernel for one workload: FETCH_SIZE and WRITE_SIZE in separate passes (kernel-trace only), written
# as profiles-style JSON to gpurun_out/r06_pmc_tsdf23_<workload>.json (copy it to profiles/: bench.py quotes it as roofline.traffic
# as long as kt_volume.hip still has the recorded hash).   usage: pmc_traffic.sh <workload> <steps>
cd /tmp && export TMPDIR=/tmp
W=${1:-farwall768}; S=${2:-6}
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmct_$c -- python $R/bench.py --workload $W --steps $S --warmup 2 --no-cpu-baseline --no-contract-ab --no-readahead --no-stress > $R/gpurun_out/pmct_$c.log 2>&1 || tail -3 $R/gpurun_out/pmct_$c.log
done
python - <<PY
import csv, glob, hashlib, json, os
out = {"kt_volume_hip_sha16": hashlib.sha256(open("$R/kintinuous_amd/csrc/kt_volume.hip", "rb").read()).hexdigest()[:16],
       "workload": "$W ($S timed frames, no read-ahead)", "kernel": "kt_tsdf23_kernel<false> / kt_tsdf23_lean_kernel<false> (whichever the run launched)",
       "collected": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, one pass each (scripts/pmc_traffic.sh), mean over the launches of the run",
       "calibration": "scripts/pmc_calibrate.sh on the kernel's own row pattern (profiles/r03_pmc_calibration_*.csv): traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob("$R/gpurun_out/pmct_%s/*/*counter_collection.csv" % c), key=os.path.getmtime)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[-1])) if ("tsdf23_kernel<false" in r["Kernel_Name"] or "tsdf23_lean_kernel<false" in r["Kernel_Name"] or "tsdf23_tol_kernel<false" in r["Kernel_Name"] or "tsdf23_sol_kernel<false" in r["Kernel_Name"]) and r["Counter_Name"] == c]
    out[c] = sum(vals) / len(vals)
    out["launches_" + c] = len(vals)
out["traffic_bytes_per_launch"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
# the algorithmic bytes of the SAME launches: the bench line of the profiled run itself (its U is counted on exactly these frames)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        line = [l for l in open("$R/gpurun_out/pmct_%s.log" % c) if l.startswith("{")][-1]
        r = json.loads(line)["roofline"]
        out["algorithmic_bytes_per_launch"] = r["algorithmic_bytes_per_launch"]
        out["U_voxels_updated"] = r["U_voxels_updated"]
        out["traffic_ratio"] = out["traffic_bytes_per_launch"] / r["algorithmic_bytes_per_launch"]
        break
    except Exception:
        pass
json.dump(out, open("$R/gpurun_out/r06_pmc_tsdf23_$W.json", "w"), indent=1)
print(json.dumps(out))
PY
