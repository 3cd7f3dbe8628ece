"""Runs one sequence through the oracle's tracker and dumps everything it produced (every pose, the voxel wraps along the way, the final
tsdf and colour volumes, every slice as a sorted point set) into an .npz.  KT_ORACLE_LIB selects the library: libkt_oracle.so (default)
or oracle/_ref/libkt_oracle_on_ref.so, the same tracker with every kernel call rerouted to the reference's own kernels
(oracle/ref_shim/kt_oracle_on_ref.h) -- tests/test_oracle_vs_ref.py::test_tracker_on_reference_kernels compares the two dumps.
usage: tracker_dump.py <icp|rgbd|rgbd_icp|dynamic_cube|static|fast_odometry> <frames> <out.npz>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from kintinuous_amd import synth
from oracle import oracle

mode, nframes, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cam = synth.Camera.small(160, 120)
N = 64
if mode in ("dynamic_cube", "static"):
    scene, traj, size = synth.Scene("room"), synth.orbit_trajectory(nframes), 6.0
else:
    scene, traj, size = synth.Scene("wall"), synth.crabwalk_trajectory(420), 7.0
frames = [synth.render(scene, cam, *traj[i]) for i in range(nframes)]
#       cols rows N fx fy cx cy size shift overlap static rgbd rgbd_icp fod dc order dynamic pr
cfg = oracle.OTrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, size, 3, 2, int(mode == "static"), int(mode == "rgbd"),
                            int(mode == "rgbd_icp"), int(mode == "fast_odometry"), 0, 0, int(mode == "dynamic_cube"), 1)
trk = oracle.OracleTracker(cfg)
poses, wraps = [], []
for k, (d, c) in enumerate(frames):
    trk.process_frame(d, c, 33333 * k)
    R, t, g = trk.pose()
    poses.append(np.concatenate([np.asarray(R, np.float32).ravel(), np.asarray(t, np.float32).ravel(), np.asarray(g, np.float32).ravel()]))
    wraps.append(np.asarray(trk.voxel_wrap(), np.int32))
trk.finalise()


def sorted_points(p):
    raw = np.ascontiguousarray(p).view(np.uint8).reshape(len(p), 32)
    key = np.concatenate([raw[:, :12], raw[:, 16:20]], axis=1)
    return key[np.lexsort(key.T[::-1])] if len(key) else key


slices = [trk.slice(i) for i in range(trk.num_slices())]
pts = [sorted_points(s[0] if isinstance(s, tuple) else s) for s in slices]
np.savez_compressed(out, poses=np.stack(poses), wraps=np.stack(wraps), volume=trk.volume(), colour=trk.color_volume(),
                    slice_sizes=np.array([len(p) for p in pts], np.int64), slice_points=np.concatenate(pts) if pts else np.zeros((0, 16), np.uint8),
                    vmap=trk.vmap_g_prev(0), nmap=trk.nmap_g_prev(0))
print(mode, "frames", nframes, "slices", len(pts), "points", int(sum(len(p) for p in pts)), "final wrap", wraps[-1].tolist())
