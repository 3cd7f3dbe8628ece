import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi, synth
cam = synth.Camera()
_, frames, traj, kw = synth.sequence("orbit", 12, cam)
ctx = abi.Ctx(0)
cfg = abi.TrackerConfig(cam.cols, cam.rows, 512, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
trk = abi.Tracker(ctx, cfg)
trk.enable_counts(True)
for k, (d, rgb) in enumerate(frames):
    trk.process_frame_host(d, rgb, k)
    U, S = trk.last_counts()
    dc = trk.debug_counts()
    print(k, "U", U, "S", S, "hopped", dc[7], "tasks", dc[2], "batches", dc[1], "in-image voxel steps", dc[3], "lane slots", dc[1] * 256, "raycast wave hop iterations", dc[5], "wave batch iterations", dc[6])
