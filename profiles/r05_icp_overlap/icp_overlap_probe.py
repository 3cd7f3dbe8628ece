"""Probe of the overlapped ICP chain (KT_ICP_OVERLAP=1): a small tracker run with a short spin bound, per-frame wall time and whether the hand-over
timed out.  Run under different GPU_MAX_HW_QUEUES to see whether hardware-queue sharing between HIP streams creates false dependencies."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C

import numpy as np

from kintinuous_amd import abi, synth

ctx = abi.Ctx(0)
abi._chk(abi.lib().kt_debug_handoff_fault(ctx.h, 0, 0, 8192, None))   # bound every spin (~2 ms)
BIG = len(sys.argv) > 1 and sys.argv[1] == "big"
READAHEAD = len(sys.argv) > 2 and sys.argv[2] == "ra"
cam = synth.Camera() if BIG else synth.Camera.small(160, 120)
scene = synth.Scene("room")
traj = synth.orbit_trajectory(300 if BIG else 12)[:12]
frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
dev = [(ctx.upload(d), ctx.upload(c)) for d, c in frames]
cfg = abi.TrackerConfig(cam.cols, cam.rows, 256 if BIG else 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
res = {}
for ov in (0, 1):
    abi._chk(abi.lib().kt_debug_icp_overlap(ov))
    trk = abi.Tracker(ctx, cfg)
    poses, times, errs = [], [], 0
    for k, (d, rgb) in enumerate(frames):
        t0 = time.perf_counter()
        if READAHEAD:
            if k + 1 < len(frames):
                trk.prefetch_frame(*dev[k + 1])
            trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
        else:
            trk.process_frame_host(d, rgb, 33333 * k)
        try:
            poses.append(np.concatenate([x.ravel() for x in trk.pose()]))
        except abi.KtError as e:
            errs += 1
            poses.append(np.zeros(15))
            trk.reset()
        times.append(1e3 * (time.perf_counter() - t0))
    res[ov] = (np.array(poses), times, errs)
    if ov:
        buf = (C.c_ulonglong * 512)()
        last = C.c_uint(0)
        abi._chk(abi.lib().kt_debug_icp_overlap_timeline(ctx.h, buf, C.byref(last)))
        T = np.array(buf, dtype=np.uint64).reshape(64, 8).astype(np.int64)
        seqs = [(last.value - 18 + i) & 63 for i in range(19)]
        t0 = T[seqs[0], 0]
        print("  iteration: wg0 in / wg255 in / pose got / timed out / published   (us after iteration 0's wg0)")
        for i, q in enumerate(seqs):
            print("  %2d: %8.2f %8.2f %8.2f %d %8.2f" % (i, (T[q, 0] - t0) / 100.0, (T[q, 1] - t0) / 100.0, (T[q, 2] - t0) / 100.0 if T[q, 2] else -1, T[q, 3], (T[q, 4] - t0) / 100.0))
    trk.close()
abi._chk(abi.lib().kt_debug_handoff_fault(ctx.h, 0, 0, 1 << 22, None))
print("big" if BIG else "small", "read-ahead" if READAHEAD else "host frames", "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "| ms/frame ordered", [round(x, 2) for x in res[0][1][2:8]], "| overlapped", [round(x, 2) for x in res[1][1][2:8]],
      "| timeouts", res[1][2], "| poses equal", bool(np.array_equal(res[0][0], res[1][0])), flush=True)
