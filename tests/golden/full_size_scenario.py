"""The bench's own configuration as one kernel-level scenario with DIGESTS as outputs (the arrays are hundreds of megabytes): two
640x480 frames of the synthetic orbit sequence integrated into a 512^3 volume with a storage wrap, the raycast from the next pose, an ICP
reduction at full resolution against that prediction, the whole volume extracted.  Run through any implementation M with the function
names of oracle/oracle.py:
   - make_full_size_ref.py runs it through oracle/_ref (the reference's own .cu sources compiled for the CPU) -> full_size_ref_v1.npz;
   - tests/test_golden_ref.py runs it through the oracle (CPU) and the HIP path (GPU) and compares the digests.
The INPUTS (three rendered frames and their poses) are stored in full_size_ref_v1.npz next to the digests: numpy's vectorised sin() and
BLAS matmul may differ in the last bit between CPUs, so regenerating them on the GPU box would not be reproducible.  NaNs are canonicalised
before hashing (the sign of a NaN is not specified on either side)."""
import hashlib

import numpy as np


def _h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def _hf(a):
    a = np.ascontiguousarray(a, np.float32).copy()
    a[np.isnan(a)] = np.float32(np.nan)     # one NaN pattern
    return _h(a.view(np.uint32) & np.where(np.isnan(a), np.uint32(0x7fffffff), np.uint32(0xffffffff)))


def _sorted_points(p):
    raw = np.ascontiguousarray(p).view(np.uint8).reshape(len(p), 32)
    key = np.concatenate([raw[:, :12], raw[:, 16:20]], axis=1)
    return key[np.lexsort(key.T[::-1])]


def make_inputs():
    """Three frames of the synthetic orbit sequence with their ground-truth poses (run where the fixture is written)."""
    from kintinuous_amd import synth
    cam = synth.Camera.scaled(1)
    _, frames, traj, _ = synth.sequence("orbit", 3, cam, 1234)
    g = {"intr": np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)}
    for k in range(3):
        g["depth%d" % k] = frames[k][0]
        g["R%d" % k] = np.asarray(traj[k][0], np.float32)
        g["c%d" % k] = np.asarray(traj[k][1], np.float32)
    g["rgb0"], g["rgb1"] = frames[0][1], frames[1][1]
    return g


class _Cam:
    pass


def scenario(M, g, intr_cls, mat33_inverse, filtered):
    """g: the stored inputs (make_inputs).  filtered: the bilateral-filtered depth frames to start from (the bilateral filter depends on
    the __expf model in the last bit of a weight; the digests are defined on the ORACLE's filtered frames, whose digest is an output)."""
    N, size = 512, 6.0
    cam = _Cam()
    cam.rows, cam.cols = g["depth0"].shape
    fx, fy, cx, cy = [float(v) for v in g["intr"]]
    frames = [(g["depth0"], g["rgb0"]), (g["depth1"], g["rgb1"]), (g["depth2"], None)]
    traj = [(g["R%d" % k], g["c%d" % k]) for k in range(3)]
    intr = intr_cls(fx, fy, cx, cy)
    trunc = max(0.06, 2.1 * size / N)
    wrap = [37, 501, 130]
    out = {"in_depth0": _h(frames[0][0]), "in_rgb1": _h(frames[1][1]), "in_filtered0": _h(filtered[0])}
    vol, col = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    pose = lambda k: (np.asarray(traj[k][0], np.float32), (np.asarray(traj[k][1], np.float32) + np.float32(size / 2)).astype(np.float32))
    for k in range(2):
        d, c = frames[k]
        Rk, tk = pose(k)
        n = M.create_nmap(M.create_vmap(intr, filtered[k]))
        M.integrate_tsdf(d, intr, [size] * 3, mat33_inverse(Rk), tk, trunc, vol, wrap, col, c, n, True)
        out["nmap%d" % k] = _hf(n)
        out["tsdf_after_%d" % k] = _h(vol)
        out["colour_after_%d" % k] = _h(col)
    Rk, tk = pose(2)
    vm, nm = np.full((3 * cam.rows, cam.cols), 7.0, np.float32), np.full((3 * cam.rows, cam.cols), -3.0, np.float32)
    cm = np.full((cam.rows, cam.cols, 4), 9, np.uint8)
    M.raycast(intr, Rk, tk, trunc, [size] * 3, vol, vm, nm, wrap, cm, col)
    out["raycast_vmap"], out["raycast_nmap"], out["raycast_bgr"] = _hf(vm), _hf(nm), _h(cm[..., :3])
    out["raycast_hits"] = int(np.isfinite(vm[:cam.rows]).sum())
    vcur = M.create_vmap(intr, filtered[2])
    ncur = M.create_nmap(vcur)
    th = float(np.sin(np.float32(20.0 * 3.14159265 / 180.0)))
    Rc = np.asarray(traj[1][0], np.float32)     # the previous frame's pose as the starting guess
    tc = pose(1)[1]
    A, b, r = M.icp_step(Rc, tc, vcur, ncur, mat33_inverse(Rk), tk, intr, vm, nm, 0.10, th)
    out["icp_A"], out["icp_b"], out["icp_residual"] = _hf(np.asarray(A, np.float32)), _hf(np.asarray(b, np.float32)), _hf(np.asarray(r, np.float32))
    pts = M.extract_cloud_slice(vol, [size] * 3, 6000000, wrap, col, 0, N, 0, N, 0, N, 1, [37, -11, 642])
    out["points"] = len(pts)
    out["points_sorted"] = _h(_sorted_points(pts))
    return out
