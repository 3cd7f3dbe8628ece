"""Several independent streams on ONE GPU (one kt_ctx + tracker each, round-robin from one host thread): how much of the GPU the
latency-bound odometry chain of a single stream leaves idle.  Not the bench configuration (that is one stream per GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi, synth
from kintinuous_amd.multistream import pingpong, stream_seed

cam = synth.Camera()
NU, STEPS, WARM = 60, 300, 20
for S in (1, 2, 3, 4):
    ctxs, trks, devs = [], [], []
    for s in range(S):
        _, frames, traj, _ = synth.sequence("orbit", NU, cam, stream_seed(s))
        ctx = abi.Ctx(0)
        cfg = abi.TrackerConfig(cam.cols, cam.rows, 512, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
        ctxs.append(ctx); trks.append(abi.Tracker(ctx, cfg))
        devs.append([(ctx.upload(d), ctx.upload(c)) for d, c in frames])

    def step(i):
        for s in range(S):
            trks[s].prefetch_frame(*devs[s][pingpong(i + 1, NU)])
            trks[s].process_frame(*devs[s][pingpong(i, NU)], 33333 * i)

    for i in range(WARM):
        step(i)
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for i in range(WARM, WARM + STEPS):
        step(i)
    for s in range(S):
        trks[s].num_poses()
        ctxs[s].sync()
    dt = time.perf_counter() - t0
    print(f"streams {S}: aggregate {S * STEPS / dt:8.1f} frames/s  ({STEPS / dt:7.1f} per stream)")
    for t in trks:
        t.close()
