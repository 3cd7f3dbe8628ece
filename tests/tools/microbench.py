"""Runs individual C-ABI kernels at VGA size repeatedly (meant to run under rocprofv3 --kernel-trace --stats)."""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from kintinuous_amd import abi, synth
from oracle import oracle
from oracle.oracle import OIntr

cam = synth.Camera()
scene = synth.Scene("room")
traj = synth.orbit_trajectory(4)
frames = [synth.render(scene, cam, R, c) for (R, c) in traj[:2]]
ctx = abi.Ctx(0)
oi = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
gi = abi.Intr(cam.fx, cam.fy, cam.cx, cam.cy)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "icp"

def maps(depth, level):
    d = oracle.bilateral_filter(depth)
    for _ in range(level):
        d = oracle.pyr_down(d)
    v = oracle.create_vmap(oi.level(level), d)
    return v, oracle.create_nmap(v)

if which == "icp":
    for level in (0, 1, 2):
        vc, nc = maps(frames[1][0], level)
        v0, n0 = maps(frames[0][0], level)
        vg, ng = oracle.transform_maps(v0, n0, np.eye(3), [3, 3, 3])
        rows, cols = vc.shape[0] // 3, vc.shape[1]
        bufs = [ctx.upload(a) for a in (vc, nc, vg, ng)]
        for _ in range(reps):
            ctx.icp_step(np.eye(3), [3, 3, 3], bufs[0], bufs[1], np.eye(3), [3, 3, 3], gi.level(level), bufs[2], bufs[3], cols, rows, 0.1, 0.342)
        print("level", level, "done")
