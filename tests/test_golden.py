"""Committed golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py from the oracle):
 - CPU: the oracle still reproduces them bit for bit (guards the checker itself);
 - GPU: the HIP path, through the C-ABI, reproduces them without touching the oracle at run time."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")


@pytest.fixture(scope="module")
def g():
    return dict(np.load(G))


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def test_oracle_reproduces_golden(oracle_mod, g):
    from oracle.oracle import OIntr, OTrackerConfig, OracleTracker
    intr = OIntr(*[float(x) for x in g["intr"]])
    f0 = oracle_mod.bilateral_filter(g["depth0"])
    assert _eq(f0, g["bilateral0"]) and _eq(oracle_mod.pyr_down(f0), g["pyr1"])
    v0 = oracle_mod.create_vmap(intr, f0)
    n0 = oracle_mod.create_nmap(v0)
    assert _eq(v0, g["vmap0"]) and _eq(n0, g["nmap0"])
    N, size, trunc = int(g["N"]), float(g["size"]), float(g["trunc"])
    vol, col = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    U, scaled = oracle_mod.integrate_tsdf(g["depth0"], intr, [size] * 3, np.eye(3), [3, 3, 3], trunc, vol, [0, 0, 0], col, g["rgb0"], n0, True)
    assert U == int(g["U"]) and _eq(scaled, g["scaled0"]) and _eq(vol, g["vol0"]) and _eq(col, g["col0"])
    vm, nm, cm = np.zeros_like(v0), np.zeros_like(v0), np.zeros((int(g["rows"]), int(g["cols"]), 4), np.uint8)
    S = oracle_mod.raycast(intr, np.eye(3), [3, 3, 3], trunc, [size] * 3, vol, vm, nm, [0, 0, 0], cm, col)
    assert S == int(g["S"]) and _eq(vm, g["ray_vmap"]) and _eq(nm, g["ray_nmap"]) and _eq(cm, g["ray_color"])
    for name, ri in (("icp", 0), ("rgbdicp", 1)):
        trk = OracleTracker(OTrackerConfig(int(g["cols"]), int(g["rows"]), N, *[float(x) for x in g["intr"]], size, 14, 2, 0, 0, ri, 0, 0, 0))
        for k in range(5):
            trk.process_frame(g[f"depth{k}"], g[f"rgb{k}"], 33333 * k)
            assert _eq(trk.dense_pose(k)[1], g[f"trk_{name}_poses"][k]), (name, k)
        assert _eq(trk.volume(), g[f"trk_{name}_vol"]) and _eq(trk.color_volume()[..., 3], g[f"trk_{name}_colw"])
        trk.close()


@pytest.mark.gpu
def test_hip_reproduces_golden(ctx, g):
    from kintinuous_amd import abi
    cols, rows = int(g["cols"]), int(g["rows"])
    intr = abi.Intr(*[float(x) for x in g["intr"]])
    d0 = ctx.upload(g["depth0"])
    f0 = ctx.empty(g["depth0"].nbytes)
    ctx.bilateral_filter(d0, f0, cols, rows)
    assert _eq(ctx.download(f0, np.uint16, (rows, cols)), g["bilateral0"])
    p1 = ctx.empty(g["pyr1"].nbytes)
    ctx.pyr_down(f0, cols, rows, p1)
    assert _eq(ctx.download(p1, np.uint16, g["pyr1"].shape), g["pyr1"])
    v0, n0 = ctx.zeros(g["vmap0"].nbytes), ctx.zeros(g["vmap0"].nbytes)
    ctx.create_vmap(intr, f0, cols, rows, v0)
    ctx.create_nmap(v0, cols, rows, n0)
    assert _eq(ctx.download(v0, np.float32, g["vmap0"].shape), g["vmap0"]) and _eq(ctx.download(n0, np.float32, g["nmap0"].shape), g["nmap0"])
    N, size, trunc = int(g["N"]), float(g["size"]), float(g["trunc"])
    vol, col = ctx.zeros(N ** 3 * 2), ctx.zeros(N ** 3 * 4)
    sc = ctx.empty(rows * cols * 4)
    ctx.integrate_tsdf(d0, cols, rows, intr, [size] * 3, np.eye(3), [3, 3, 3], trunc, vol, sc, [0, 0, 0], col, ctx.upload(g["rgb0"]), n0, True, N)
    assert _eq(ctx.download(sc, np.float32, (rows, cols)), g["scaled0"])
    assert _eq(ctx.download(vol, np.int16, (N, N, N)), g["vol0"]) and _eq(ctx.download(col, np.uint8, (N, N, N, 4)), g["col0"])
    vm, nm, cm = ctx.zeros(g["vmap0"].nbytes), ctx.zeros(g["vmap0"].nbytes), ctx.zeros(rows * cols * 4)
    ctx.raycast(intr, np.eye(3), [3, 3, 3], trunc, [size] * 3, vol, vm, nm, cols, rows, [0, 0, 0], cm, col, N)
    assert _eq(ctx.download(vm, np.float32, g["ray_vmap"].shape), g["ray_vmap"]) and _eq(ctx.download(nm, np.float32, g["ray_nmap"].shape), g["ray_nmap"])
    assert _eq(ctx.download(cm, np.uint8, (rows, cols, 4)), g["ray_color"])
    cap = 100000
    out = ctx.empty(cap * 32)
    n = ctx.extract_cloud_slice(vol, [size] * 3, out, cap, [0, 0, 0], col, 0, N, 0, N, 0, N, 1, [0, 0, 0], N)
    pts = ctx.download(out, abi.POINT_DTYPE, (cap,))[:n]
    order = np.lexsort((pts["xyz"][:, 2], pts["xyz"][:, 1], pts["xyz"][:, 0]))
    assert _eq(pts["xyz"][order], g["cloud_xyz"]) and _eq(pts["bgra"][order], g["cloud_bgra"])
    # ICP system of frame 1 vs frame 0 (reference-order sums: bit-exact)
    d1 = ctx.upload(g["depth1"])
    f1, v1, n1 = ctx.empty(g["depth1"].nbytes), ctx.zeros(g["vmap0"].nbytes), ctx.zeros(g["vmap0"].nbytes)
    ctx.bilateral_filter(d1, f1, cols, rows)
    ctx.create_vmap(intr, f1, cols, rows, v1)
    ctx.create_nmap(v1, cols, rows, n1)
    vg, ng = ctx.zeros(g["vmap0"].nbytes), ctx.zeros(g["vmap0"].nbytes)
    ctx.transform_maps(v0, n0, cols, rows, np.eye(3), [3, 3, 3], vg, ng)
    A, b, r = ctx.icp_step(np.eye(3), [3, 3, 3], v1, n1, np.eye(3), [3, 3, 3], intr, vg, ng, cols, rows, 0.10, float(g["angle_thres"]))
    assert _eq(A, g["icp_A"]) and _eq(b, g["icp_b"]) and _eq(r, g["icp_r"])
    # whole-frame tracker
    for name, ri in (("icp", 0), ("rgbdicp", 1)):
        cfg = abi.TrackerConfig(cols, rows, N, *[float(x) for x in g["intr"]], size, 14, 2, 0, 0, ri, 0, 0, 0)
        trk = abi.Tracker(ctx, cfg)
        for k in range(5):
            trk.process_frame_host(g[f"depth{k}"], g[f"rgb{k}"], 33333 * k)
            assert np.abs(trk.dense_pose(k)[1] - g[f"trk_{name}_poses"][k]).max() < 1e-6, (name, k)
        assert _eq(trk.volume(), g[f"trk_{name}_vol"]) and _eq(trk.color_volume()[..., 3], g[f"trk_{name}_colw"])
        trk.close()


# ---- golden_v2: full colour volumes, -p ground truth, -d dynamic cube, view products, JPEG colour -------------------------------------
G2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v2.npz")


@pytest.fixture(scope="module")
def g2():
    return dict(np.load(G2))


def _cfg_args(g, **kw):
    d = dict(voxel_shift=14, dynamic_cube=0)
    d.update(kw)
    return (int(g["cols"]), int(g["rows"]), 64, *[float(x) for x in g["intr"]], 6.0, d["voxel_shift"], 2, 0, 0, 0, 0, 0, 0, d["dynamic_cube"])


def _run_v2(g2, make, finish):
    """The three tracker scenarios of golden_v2 on a tracker factory `make(args)`; `finish(trk)` closes it."""
    out = {}
    trk = make(_cfg_args(g2))
    for k in range(4):
        trk.process_frame(g2[f"depth{k}"], g2[f"rgb{k}"], 33333 * k)
    out["icp_vol"], out["icp_color"] = trk.volume().copy(), trk.color_volume().copy()
    finish(trk)
    trk = make(_cfg_args(g2))
    trk.load_trajectory(g2["gt_stamps"], g2["gt_rows"])
    for k in range(6):
        trk.process_frame(g2[f"depth{k}"], g2[f"rgb{k}"], int(g2["gt_all_stamps"][k]))
    out["gt_poses"] = np.stack([trk.dense_pose(i)[1] for i in range(trk.num_poses())])
    out["gt_vol"], out["gt_color"] = trk.volume().copy(), trk.color_volume().copy()
    finish(trk)
    trk = make(_cfg_args(g2, dynamic_cube=1, voxel_shift=2))
    basis, wraps = [], []
    for k in range(8):
        trk.process_frame(g2[f"dyn_depth{k}"], g2[f"dyn_rgb{k}"], 33333 * k)
        basis.append(trk.volume_basis())
        wraps.append(np.array(trk.voxel_wrap()))
    out["dyn_basis"], out["dyn_wrap"] = np.stack(basis), np.stack(wraps)
    out["dyn_poses"] = np.stack([trk.dense_pose(i)[1] for i in range(trk.num_poses())])
    out["dyn_vol"], out["dyn_color"] = trk.volume().copy(), trk.color_volume().copy()
    finish(trk)
    return out


def test_oracle_reproduces_golden_v2(oracle_mod, g2):
    from oracle.oracle import OTrackerConfig, OracleTracker
    got = _run_v2(g2, lambda a: OracleTracker(OTrackerConfig(*a)), lambda t: t.close())
    for key, val in got.items():
        assert _eq(val, g2[key]), key
    img, col = oracle_mod.generate_image(g2["view_vmap"], g2["view_nmap"], g2["view_vcol"], [-18.0, -18.0, -18.0])
    assert _eq(img, g2["view_img"]) and _eq(col, g2["view_color"])
    assert _eq(oracle_mod.generate_depth(oracle_mod.mat33_inverse(g2["view_R"]), g2["view_t"], g2["view_vmap"], g2["view_nmap"]), g2["view_depth"])


def test_jpeg_decoders_reproduce_golden_v2(g2, tmp_path):
    """Both the numpy reference decoder and the C++ decoder of the .klg reader give the committed bytes for the committed streams."""
    import subprocess
    from kintinuous_amd import build, jpeg_ref
    build.build_host()
    cols, rows = int(g2["cols"]), int(g2["rows"])
    for name in ("420", "422r", "444n"):
        data = g2[f"jpeg_{name}"].tobytes()
        assert _eq(jpeg_ref.decode(data), g2[f"jpeg_{name}_bgr"]), name
        src, dst = tmp_path / f"{name}.jpg", tmp_path / f"{name}.bgr"
        src.write_bytes(data)
        r = subprocess.run([build.JPEG_TOOL, str(src), str(cols), str(rows), str(dst)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        assert _eq(np.frombuffer(dst.read_bytes(), np.uint8).reshape(rows, cols, 3), g2[f"jpeg_{name}_bgr"]), name


@pytest.mark.gpu
def test_hip_reproduces_golden_v2(ctx, g2):
    from kintinuous_amd import abi

    class Host:   # the tracker fed with host frames, with the oracle tracker's method names
        def __init__(self, args):
            self.t = abi.Tracker(ctx, abi.TrackerConfig(*args))

        def process_frame(self, d, rgb, ts):
            self.t.process_frame_host(d, rgb, ts)

        def __getattr__(self, name):
            return getattr(self.t, name)

    got = _run_v2(g2, Host, lambda t: t.t.close())
    for key, val in got.items():
        if key.endswith("_poses") and key != "gt_poses":
            assert np.abs(val - g2[key]).max() < 1e-6, key      # device libm in Rodrigues: see test_gpu_tracker.py
        else:
            assert _eq(val, g2[key]), key
    rows, cols = int(g2["rows"]), int(g2["cols"])
    dv, dn, dc = ctx.upload(g2["view_vmap"]), ctx.upload(g2["view_nmap"]), ctx.upload(g2["view_vcol"])
    di, dcol, dd = ctx.empty(rows * cols * 3), ctx.empty(rows * cols * 3), ctx.empty(rows * cols * 2)
    ctx.generate_image(dv, dn, dc, cols, rows, [-18.0, -18.0, -18.0], 1, di, dcol)
    ctx.generate_depth(abi.host_mat33_inverse(g2["view_R"]), g2["view_t"], dv, dn, cols, rows, dd)
    assert _eq(ctx.download(di, np.uint8, (rows, cols, 3)), g2["view_img"]) and _eq(ctx.download(dcol, np.uint8, (rows, cols, 3)), g2["view_color"])
    assert _eq(ctx.download(dd, np.uint16, (rows, cols)), g2["view_depth"])
