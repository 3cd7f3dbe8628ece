"""Generates tests/golden/golden_v1.npz: seeded synthetic inputs and the oracle's outputs for every hot-path function.

The reference has no fixtures of its own and cannot run here, so these vectors are dumps of the CPU
restatement (since round 2 the restatement itself is pinned against the reference's kernel sources,
tests/test_oracle_vs_ref.py, and golden_ref_v1.npz holds outputs of that build): they pin the oracle against silent drift (compiler, flags, edits) and give the GPU tests
a committed known answer that does not depend on rebuilding the oracle.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "4")

from kintinuous_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle.oracle import OIntr, OTrackerConfig, OracleTracker  # noqa: E402


def main():
    cam = synth.Camera.small(64, 48)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(300)[::6][:5]  # 5 frames, ~6 frame-steps apart (a few cm) so ICP has work to do
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    g = {"cols": cam.cols, "rows": cam.rows, "intr": np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float64)}
    for k, (d, rgb) in enumerate(frames):
        g[f"depth{k}"], g[f"rgb{k}"] = d, rgb
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    d0, rgb0 = frames[0]
    f0 = oracle.bilateral_filter(d0)
    g["bilateral0"] = f0
    g["pyr1"] = oracle.pyr_down(f0)
    v0 = oracle.create_vmap(intr, f0)
    n0 = oracle.create_nmap(v0)
    g["vmap0"], g["nmap0"] = v0, n0
    N, size = 32, 6.0
    trunc = max(0.06, 2.1 * size / N)
    vol = np.zeros((N, N, N), np.int16)
    col = np.zeros((N, N, N, 4), np.uint8)
    U, scaled = oracle.integrate_tsdf(d0, intr, [size] * 3, np.eye(3), [3, 3, 3], trunc, vol, [0, 0, 0], col, rgb0, n0, True)
    g["N"], g["size"], g["trunc"], g["U"] = N, size, np.float32(trunc), U
    g["scaled0"], g["vol0"], g["col0"] = scaled, vol.copy(), col.copy()
    vm, nm, cm = np.zeros_like(v0), np.zeros_like(v0), np.zeros((cam.rows, cam.cols, 4), np.uint8)
    g["S"] = oracle.raycast(intr, np.eye(3), [3, 3, 3], trunc, [size] * 3, vol, vm, nm, [0, 0, 0], cm, col)
    g["ray_vmap"], g["ray_nmap"], g["ray_color"] = vm, nm, cm
    pts = oracle.extract_cloud_slice(vol, [size] * 3, 100000, [0, 0, 0], col, 0, N, 0, N, 0, N, 1, [0, 0, 0])
    order = np.lexsort((pts["xyz"][:, 2], pts["xyz"][:, 1], pts["xyz"][:, 0]))
    g["cloud_xyz"], g["cloud_bgra"] = pts["xyz"][order], pts["bgra"][order]
    # ICP system of frame 1 against frame 0
    f1 = oracle.bilateral_filter(frames[1][0])
    v1 = oracle.create_vmap(intr, f1)
    n1 = oracle.create_nmap(v1)
    vg, ng = oracle.transform_maps(v0, n0, np.eye(3), [3, 3, 3])
    ang = float(np.float32(np.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))
    A, b, r = oracle.icp_step(np.eye(3), [3, 3, 3], v1, n1, np.eye(3), [3, 3, 3], intr, vg, ng, 0.10, ang, order=0)
    g["icp_A"], g["icp_b"], g["icp_r"], g["angle_thres"] = A, b, r, np.float32(ang)
    # RGB-D pieces
    dm0, dm1 = oracle.depth_to_metres(d0, 6000), oracle.depth_to_metres(frames[1][0], 6000)
    i0, i1 = oracle.bgr_to_intensity(rgb0), oracle.bgr_to_intensity(frames[1][1])
    dx, dy = oracle.derivative_images(i1)
    g["metres0"], g["intensity0"], g["dIdx1"], g["dIdy1"] = dm0, i0, dx, dy
    g["gauss_f32"], g["gauss_u8"] = oracle.pyr_down_gauss_f32(dm0), oracle.pyr_down_gauss_u8(i0)
    K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]])
    corres, sigma, count = oracle.rgb_residual(np.float32(12.0 ** 2 / 0.125 ** 2), dx, dy, dm0, dm1, i0, i1, np.float32(0.07), np.zeros(3, np.float32),
                                               np.eye(3, dtype=np.float32))
    g["rgb_sigma"], g["rgb_count"], g["rgb_valid"] = sigma, count, corres["valid"]
    cloud = oracle.project_to_cloud(dm0, cam.fx, cam.fy, cam.cx, cam.cy, 0)
    Ar, br = oracle.rgb_step(corres, float(np.sqrt(np.float32(max(count, 1)))), cloud, np.float32(cam.fx), np.float32(cam.fy), dx, dy, 0.125, order=0)
    g["rgb_A"], g["rgb_b"] = Ar, br
    # whole-frame tracker over the 5 frames, ICP and ICP + RGB-D
    for name, ri in (("icp", 0), ("rgbdicp", 1)):
        trk = OracleTracker(OTrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, size, 14, 2, 0, 0, ri, 0, 0, 0))
        poses = []
        for k, (d, rgb) in enumerate(frames):
            trk.process_frame(d, rgb, 33333 * k)
            poses.append(trk.dense_pose(k)[1])
        g[f"trk_{name}_poses"] = np.stack(poses)
        g[f"trk_{name}_vol"] = trk.volume().copy()
        g[f"trk_{name}_colw"] = trk.color_volume()[..., 3].copy()
        trk.close()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
