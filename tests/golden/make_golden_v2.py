"""Generates tests/golden/golden_v2.npz: known answers for the pieces added after golden_v1 -- the full colour volume of the
tracker runs (r, g, b and weight), ground-truth odometry (-p) with a dropped frame, the dynamic cube (-d), the view products
(generateImage / generateDepth) and the .klg JPEG colour path (a stream from kintinuous_amd/jpeg_ref.py with its decoded bytes).
Same status as golden_v1: dumps of the CPU restatement (see make_golden.py).   Run:  python tests/golden/make_golden_v2.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "4")

from kintinuous_amd import jpeg_ref, synth  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle.oracle import OTrackerConfig, OracleTracker  # noqa: E402
from scipy.spatial.transform import Rotation  # noqa: E402

N, SIZE = 64, 6.0      # at 64x48 pixels coarser volumes lose track within a few frames


def cfg(cam, **kw):
    d = dict(voxel_shift=14, static_mode=0, use_rgbd=0, use_rgbd_icp=0, dynamic_cube=0)
    d.update(kw)
    return (cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, SIZE, d["voxel_shift"], 2, d["static_mode"], d["use_rgbd"], d["use_rgbd_icp"],
            0, 0, 0, d["dynamic_cube"])


def main():
    cam = synth.Camera.small(64, 48)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(300)[::2][:6]
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    g = {"cols": cam.cols, "rows": cam.rows, "intr": np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float64)}
    for k, (d, rgb) in enumerate(frames):
        g[f"depth{k}"], g[f"rgb{k}"] = d, rgb
    # ICP run (4 frames: the tiny image keeps track that long): full colour volume
    trk = OracleTracker(OTrackerConfig(*cfg(cam)))
    for k, (d, rgb) in enumerate(frames[:4]):
        trk.process_frame(d, rgb, 33333 * k)
    assert np.abs(trk.pose()[1] - (traj[3][1] + 3)).max() < 0.05
    g["icp_color"] = trk.color_volume().copy()
    g["icp_vol"] = trk.volume().copy()
    trk.close()
    # ground truth (-p): frame 3 has no trajectory entry
    rows = synth.ground_truth_rows(traj)
    stamps = np.array([1000 * (k + 1) for k in range(len(frames))], np.uint64)
    keep = [0, 1, 2, 4, 5]
    g["gt_stamps"], g["gt_rows"], g["gt_all_stamps"] = stamps[keep], rows[keep], stamps
    trk = OracleTracker(OTrackerConfig(*cfg(cam)))
    trk.load_trajectory(stamps[keep], rows[keep])
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame(d, rgb, int(stamps[k]))
    g["gt_poses"] = np.stack([trk.dense_pose(i)[1] for i in range(trk.num_poses())])
    g["gt_vol"], g["gt_color"] = trk.volume().copy(), trk.color_volume().copy()
    # view products of that model (the raycast colour image is not exposed by the oracle tracker: a fixed pseudo-random one stands in)
    vcol = np.random.default_rng(4).integers(0, 256, (cam.rows, cam.cols, 4), dtype=np.uint8)
    R, t, _ = trk.pose()
    v, n = trk.vmap_g_prev(0).copy(), trk.nmap_g_prev(0).copy()
    assert np.isfinite(v[: cam.rows]).mean() > 0.5, np.isfinite(v[: cam.rows]).mean()
    g["view_vmap"], g["view_nmap"], g["view_vcol"], g["view_R"], g["view_t"] = v, n, vcol, R, t
    g["view_img"], g["view_color"] = oracle.generate_image(v, n, vcol, [-18.0, -18.0, -18.0])
    g["view_depth"] = oracle.generate_depth(oracle.mat33_inverse(R), t, v, n)
    trk.close()
    # dynamic cube (-d): the camera turns on the spot, shift threshold 2 voxels
    yaws = [0.03 * k for k in range(8)]
    dframes = [synth.render(scene, cam, Rotation.from_euler("y", a).as_matrix(), np.zeros(3)) for a in yaws]
    for k, (d, rgb) in enumerate(dframes):
        g[f"dyn_depth{k}"], g[f"dyn_rgb{k}"] = d, rgb
    trk = OracleTracker(OTrackerConfig(*cfg(cam, dynamic_cube=1, voxel_shift=2)))
    basis, wraps = [], []
    for k, (d, rgb) in enumerate(dframes):
        trk.process_frame(d, rgb, 33333 * k)
        basis.append(trk.volume_basis())
        wraps.append(np.array(trk.voxel_wrap()))
    g["dyn_basis"], g["dyn_wrap"] = np.stack(basis), np.stack(wraps)
    g["dyn_poses"] = np.stack([trk.dense_pose(i)[1] for i in range(trk.num_poses())])
    g["dyn_vol"], g["dyn_color"] = trk.volume().copy(), trk.color_volume().copy()
    trk.close()
    assert np.abs(g["dyn_wrap"]).max() > 0 and not np.array_equal(g["dyn_basis"][0], g["dyn_basis"][-1])
    # JPEG colour: three stream layouts of frame 0's image, with the decoded bytes
    img = np.ascontiguousarray(frames[0][1])
    for name, kw in (("420", dict(subsampling="420")), ("422r", dict(subsampling="422", restart_interval=2, ac_table="skewed")),
                     ("444n", dict(subsampling="444", interleaved=False))):
        data = jpeg_ref.encode(img, quality=88, **kw)
        g[f"jpeg_{name}"] = np.frombuffer(data, np.uint8)
        g[f"jpeg_{name}_bgr"] = jpeg_ref.decode(data)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v2.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
