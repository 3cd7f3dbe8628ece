"""Generates tests/golden/golden_ref_v1.npz: seeded inputs and the outputs of THE REFERENCE'S OWN kernel sources
(/root/reference/src/frontend/cuda/*.cu, compiled for the CPU as oracle/_ref/libkt_ref.so by oracle/Makefile) for the scenario in
ref_scenario.py.  Unlike golden_v1 / golden_v2 (dumps of the restatement), these vectors come from the reference's code;
/root/reference does not exist on the GPU box, so they are committed.   Run (build container only):  python tests/golden/make_golden_ref.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from kintinuous_amd import synth  # noqa: E402
from oracle import oracle, ref  # noqa: E402
from oracle.oracle import OIntr  # noqa: E402
from ref_scenario import scenario  # noqa: E402


def inputs():
    cam = synth.Camera.small(96, 72)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(300)[::6][:2]
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    g = {"cols": cam.cols, "rows": cam.rows, "intr": np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float64)}
    rng = np.random.default_rng(2026)
    for k, (d, rgb) in enumerate(frames):
        d = d.copy()
        d[rng.random(d.shape) < 0.01] = 0        # holes
        g[f"depth{k}"], g[f"rgb{k}"] = d, rgb
    N, size = 48, 6.0
    g["N"], g["size"], g["trunc"] = N, size, np.float32(max(0.06, 2.1 * size / N))
    g["wrap"], g["real_wrap"] = np.array([5, 44, 17], np.int32), np.array([53, -4, 17], np.int32)
    for k, (R, c) in enumerate(traj):
        Rk = np.asarray(R, np.float32)
        g[f"R{k}"], g[f"t{k}"] = Rk, (np.asarray(c, np.float32) + np.float32(3)).astype(np.float32)
        g[f"Rinv{k}"] = oracle.mat33_inverse(Rk)     # host-side Eigen inverse: restated by the oracle, an input here
    g["R_g"], g["t_g"] = oracle.rodrigues(np.array([0.1, 0.2, -0.05])).astype(np.float32), np.array([0.1, -0.2, 0.3], np.float32)
    g["angle_thres"] = np.float32(np.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0)))
    K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]])
    g["krkinv"] = (K @ oracle.rodrigues(np.array([0.002, -0.003, 0.001])) @ np.linalg.inv(K)).astype(np.float32)
    g["kt"] = (K @ np.array([0.004, -0.002, 0.003])).astype(np.float32)
    g["min_scale"] = np.float32(5.0 ** 2 / 0.125 ** 2)
    return g


def main():
    g = inputs()
    ref.build()
    g["filtered0"], g["filtered1"] = ref.bilateral_filter(g["depth0"]), ref.bilateral_filter(g["depth1"])
    out = scenario(ref, g, OIntr)
    assert int(out["rgb_sigma_count"][1]) > 50 and len(out["cloud"]) > 300 and np.isfinite(out["ray_vmap"][: int(g["rows"])]).sum() > 500
    assert float(out["icp_r"][1]) > 1000
    # the z + 1 wrap of the last plane is exercised: the rotated volume has a wall's worth of crossings between logical N - 1 and 0, and
    # the extraction of its top slab holds at least that many points more than a wrap-less extraction could
    assert int(out["zwrap_shift_crossings"][1]) > 100 and len(out["cloud_zwrap_top"]) >= int(out["zwrap_shift_crossings"][1])
    path = os.path.join(HERE, "golden_ref_v1.npz")
    np.savez_compressed(path, **{f"in_{k}": v for k, v in g.items()}, **{f"out_{k}": v for k, v in out.items()})
    print(path, os.path.getsize(path), "bytes;", len(out), "outputs")


if __name__ == "__main__":
    main()
