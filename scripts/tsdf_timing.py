"""In-kernel timeline of kt_tsdf23_kernel: apply scripts/tsdf_timing.patch (git apply), build with KT_EXTRA_FLAGS=-DKT_TSDF_TIMING, run this
on the GPU box, then revert the patch (it is kept out of kt_volume.hip so that the file hash the PMC traffic files record stays valid).
Per wave, 100 MHz ticks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi, synth
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "orbit512"
cfg_name, scale, N, kw = bench.WORKLOADS[w]
cam = synth.Camera.scaled(scale)
_, frames, traj, kw2 = synth.sequence(cfg_name, 8, cam, 1234)
d = dict(volume_size=6.0, voxel_shift=14, overlap=2, static_mode=0, use_rgbd=0, use_rgbd_icp=0, fast_odometry=0, disable_color_angle=0)
d.update(kw2); d.update(kw)
ctx = abi.Ctx(0)
cfg = abi.TrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, d["volume_size"], d["voxel_shift"], d["overlap"], d["static_mode"],
                        d["use_rgbd"], d["use_rgbd_icp"], d["fast_odometry"], d["disable_color_angle"], 0)
trk = abi.Tracker(ctx, cfg)
for i, (dep, rgb) in enumerate(frames):
    trk.process_frame(ctx.upload(dep), ctx.upload(rgb), 33333 * i)
    trk.num_poses()
ctx.sync()
out = np.zeros(8192 * 8, np.uint64)
f = abi.lib().kt_debug_tsdf_timing
f.argtypes = [C.c_void_p]; f.restype = C.c_int
assert f(out.ctypes.data_as(C.c_void_p)) == 0
t = out.reshape(8192, 8).astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) / 100.0   # microseconds
print("kernel entry  (us after the first wave): p50 %.2f p99 %.2f max %.2f" % tuple(np.percentile(rel[:, 0], [50, 99, 100])))
has = t[:, 1] > 0
print("waves with a task: %d of 8192" % has.sum())
r = rel[has]
print("first task set up: p50 %.2f p99 %.2f" % tuple(np.percentile(r[:, 1] - r[:, 0], [50, 99])))
for k in range(2, 6):
    ok = t[has][:, k] > t[has][:, k - 1]
    print("batch %d: p50 %.2f p99 %.2f  (n=%d)" % ((k - 1,) + tuple(np.percentile((r[:, k] - r[:, k - 1])[ok], [50, 99])) + (ok.sum(),)))
print("exit: p50 %.2f p99 %.2f max %.2f" % tuple(np.percentile(rel[:, 6], [50, 99, 100])))
hw = (out.reshape(8192, 8)[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (out.reshape(8192, 8)[:, 7] >> np.uint64(32)).astype(np.int64) & 0xf
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
end = rel[:, 6]
busy = np.where(has, rel[:, 6] - rel[:, 0], 0.0)
import collections
fin = collections.defaultdict(float); work = collections.defaultdict(float); cnt = collections.Counter()
for k, e, b in zip(key, end, busy):
    fin[k] = max(fin[k], e); work[k] += b; cnt[k] += 1
f = np.array(list(fin.values())); wk = np.array([work[k] for k in fin]); c = np.array([cnt[k] for k in fin])
print("SIMDs seen: %d, waves per SIMD min/max %d/%d" % (len(f), c.min(), c.max()))
print("SIMD finish time: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(f, [10, 50, 90, 100])))
print("sum of wave lifetimes per SIMD: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(wk, [10, 50, 90, 100])))
cukey = key // 4
cf = collections.defaultdict(float)
for k, e in zip(cukey, end): cf[k] = max(cf[k], e)
cfv = np.array(list(cf.values()))
print("CUs seen: %d; CU finish time p10 %.1f p50 %.1f p90 %.1f max %.1f" % ((len(cfv),) + tuple(np.percentile(cfv, [10, 50, 90, 100]))))
xf = collections.defaultdict(float)
for k, e in zip(xcc, end): xf[k] = max(xf[k], e)
print("XCD finish:", {int(k): round(v, 1) for k, v in sorted(xf.items())})
# the slowest waves: which task stage dominates
slow = np.argsort(-end)[:12]
for wv in slow:
    print("wave %5d xcc %d cu %2d simd %d: entry %.1f setup %.1f batches %s end %.1f" % (wv, xcc[wv], cu[wv], simd[wv], rel[wv, 0], rel[wv, 1] - rel[wv, 0],
          np.round(np.diff(rel[wv, 1:6]), 1), rel[wv, 6]))
