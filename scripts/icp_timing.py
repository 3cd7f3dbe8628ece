import os, sys, ctypes as C
os.environ.setdefault("KT_ICP_LEVELS", "0")   # the probes live in kt_icp_kernel (one launch per iteration), not in the level kernel
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi, synth
cam = synth.Camera()
_, frames, traj, kw = synth.sequence("orbit", 6, cam)
ctx = abi.Ctx(0)
ri = len(sys.argv) > 1 and sys.argv[1] == "-ri"   # the joint RGB-D + ICP kernel's probes instead of the ICP kernel's
cfg = abi.TrackerConfig(cam.cols, cam.rows, 512, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 1 if ri else 0, 1 if ri else 0, 0, 0, 0)
trk = abi.Tracker(ctx, cfg)
for k, (d, rgb) in enumerate(frames):
    trk.process_frame_host(d, rgb, k)
# state_dev is private; read icp29 of the last iteration (level 0) through the pinned host mirror offset: use debug hook
print(("joint kernel: " if ri else "") + "ticks(10ns) from the start of the sweeping block: loop, publish, (same), sweep+fold, ldlt, pose update, state stored (joint: after K R K^-1):", trk.debug_state()[:8])

# when the workgroups of that launch entered / left their pixel loops and published, relative to the sweeping workgroup's entry
buf = (C.c_ulonglong * 768)()
if abi.lib().kt_debug_icp_wg_times(ctx.h, buf) == 0:
    T = np.frombuffer(buf, dtype=np.uint64).reshape(3, 256).astype(np.int64)
    t0 = T[0, 255]
    pct = lambda a: "p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(a, [10, 50, 90, 100]))
    print("workgroup entry - sweeper's entry (us):", pct((T[0] - t0) * 0.01), "| first", (T[0].min() - t0) * 0.01)
    print("loop done - sweeper's entry:", pct((T[1] - t0) * 0.01), "| sweeper", (T[1, 255] - t0) * 0.01)
    print("published - sweeper's entry:", pct((T[2] - t0) * 0.01), "| sweeper", (T[2, 255] - t0) * 0.01)
    print("loop length per workgroup:", pct((T[1] - T[0]) * 0.01))
    late = np.argsort(T[2])[-8:]
    print("last publishers (workgroup, us):", [(int(w), round((T[2, w] - t0) * 0.01, 2)) for w in late])
