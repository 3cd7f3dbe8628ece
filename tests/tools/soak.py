"""Soak: long shifting sequence, HIP tracker (device frames + read-ahead) vs the oracle tracker, every pose compared.
usage: soak.py [icp|rgbd_icp|rgbd] [cols=160] [N=96]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np
from kintinuous_amd import abi, synth
from oracle import oracle
COLS = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cam = synth.Camera.small(COLS, COLS * 3 // 4)
scene = synth.Scene("wall")
traj = synth.crabwalk_trajectory(420)
frames = [synth.render(scene, cam, *traj[i]) for i in range(0, 420, 1)]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 96
MODE = sys.argv[1] if len(sys.argv) > 1 else "icp"          # icp | rgbd_icp | rgbd
if MODE != "icp":
    frames = frames[:160]                                   # the RGB-D oracle is slower; 160 frames still shift the volume 10 times
args = (cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 3, 2, 0, int(MODE == "rgbd"), int(MODE == "rgbd_icp"), 0, 0, 0)
ctx = abi.Ctx(0)
trk, otr = abi.Tracker(ctx, abi.TrackerConfig(*args)), oracle.OracleTracker(oracle.OTrackerConfig(*args))
dev = [(ctx.upload(d), ctx.upload(c)) for d, c in frames]
worst = 0.0
t0 = time.time()
for k in range(len(frames)):
    trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
    if k + 1 < len(frames):
        trk.prefetch_frame(*dev[k + 1])
    otr.process_frame(frames[k][0], frames[k][1], 33333 * k)
    R, t, gc = trk.pose()
    Ro, to, go = otr.pose()
    worst = max(worst, float(np.abs(R - Ro).max()), float(np.abs(gc - go).max()))
    assert np.array_equal(trk.voxel_wrap(), otr.voxel_wrap()), k
assert trk.num_slices() == otr.num_slices()
v, ov = trk.volume(), otr.volume()
print(MODE, "frames", len(frames), "slices", trk.num_slices(), "worst pose diff", worst, "tsdf mismatches", int((v != ov).sum()), "of", int((otr.color_volume()[..., 3] != 0).sum()),
      "time %.1fs" % (time.time() - t0))
c, oc = trk.color_volume(), otr.color_volume()
print("colour mismatches per channel", [int((c[..., ch] != oc[..., ch]).sum()) for ch in range(4)])
assert worst < 1e-5 and np.array_equal(v, ov) and np.array_equal(c, oc)
