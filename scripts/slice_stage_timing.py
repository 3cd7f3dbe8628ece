"""Time of the device-resident slice stage (kt_slice_process_device: weight cull, voxel grid at one voxel, 20-NN normals) per 100 000 input
points, on slabs a crab-walk's tracker extracts (640x480 into 512^3, 7 m): the stage is enqueued on its workspace's stream and the
host waits once per call.   python scripts/slice_stage_timing.py > gpurun_out/r03_slice_stage_timing.md"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi, synth

ctx = abi.Ctx(0)
cam = synth.Camera.scaled(1)
_, frames, _, kw = synth.sequence("crabwalk", 60, cam, 1234)
cfg = abi.TrackerConfig(cam.cols, cam.rows, 512, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 14, 2, 0, 0, 0, 0, 0, 0)
trk = abi.Tracker(ctx, cfg)
for k, (d, rgb) in enumerate(frames):
    trk.process_frame_host(d, rgb, 33333 * k)
trk.finalise()
slices = [trk.slice(i)[0] for i in range(trk.num_slices())]
trk.close()
leaf = 7.0 / 512
ws = C.c_void_p()
cap = max(len(s) for s in slices) + 16
abi._chk(abi.lib().kt_slice_ws_create(ctx.h, cap, None, C.byref(ws)))
print("| slice | input points | output points | ms per call | us per 100 k input points |")
print("|---|---|---|---|---|")
for i, s in enumerate(slices):
    if len(s) < 1000:
        continue
    pts = ctx.upload(s)
    n_dev = ctx.upload(np.array([len(s)], np.uint32))
    n = C.c_size_t(0)
    for rep in range(3):   # warm
        abi._chk(abi.lib().kt_slice_process_device(ws, pts.ptr, n_dev.ptr, len(s), 8, leaf, 20))
        abi._chk(abi.lib().kt_slice_ws_count(ws, C.byref(n)))
    R = 10
    t0 = time.perf_counter()
    for rep in range(R):
        abi._chk(abi.lib().kt_slice_process_device(ws, pts.ptr, n_dev.ptr, len(s), 8, leaf, 20))
        abi._chk(abi.lib().kt_slice_ws_count(ws, C.byref(n)))
    ms = 1e3 * (time.perf_counter() - t0) / R
    print("| %d | %d | %d | %.3f | %.0f |" % (i, len(s), n.value, ms, 1e3 * ms / (len(s) / 1e5)))
abi._chk(abi.lib().kt_slice_ws_destroy(ws))
