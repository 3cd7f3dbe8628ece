import os, sys, ctypes as C
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
