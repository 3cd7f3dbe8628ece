"""GPU parity: cyclical TSDF volume kernels through the C-ABI vs the oracle.
Bar: bit-exact on every integer / byte / index output and on the float maps (same IEEE operation order)."""
import numpy as np
import pytest

from conftest import random_rotation

pytestmark = pytest.mark.gpu


def _maps(oracle, cam, depth):
    from oracle.oracle import OIntr
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    vmap = oracle.create_vmap(intr, oracle.bilateral_filter(depth))
    nmap = oracle.create_nmap(vmap)
    return intr, vmap, nmap


def _integrate_both(ctx, oracle, cam, depth, rgb, nmap, N, size, Rinv, t, trunc, wrap, angle, vol0=None, col0=None):
    from kintinuous_amd.abi import Intr
    from oracle.oracle import OIntr
    vol = np.zeros((N, N, N), np.int16) if vol0 is None else vol0.copy()
    col = np.zeros((N, N, N, 4), np.uint8) if col0 is None else col0.copy()
    dvol, dcol = ctx.upload(vol), ctx.upload(col)
    U, scaled = oracle.integrate_tsdf(depth, OIntr(cam.fx, cam.fy, cam.cx, cam.cy), [size] * 3, Rinv, t, trunc, vol, wrap, col, rgb, nmap, angle)
    dscaled = ctx.empty(depth.size * 4)
    ctx.integrate_tsdf(ctx.upload(depth), cam.cols, cam.rows, Intr(cam.fx, cam.fy, cam.cx, cam.cy), [size] * 3, Rinv, t, trunc, dvol, dscaled,
                       wrap, dcol, ctx.upload(rgb), ctx.upload(nmap), angle, N)
    ctx.sync()
    gvol = ctx.download(dvol, np.int16, (N, N, N))
    gcol = ctx.download(dcol, np.uint8, (N, N, N, 4))
    gscaled = ctx.download(dscaled, np.float32, depth.shape)
    return (vol, col, scaled, U), (gvol, gcol, gscaled), (dvol, dcol)


@pytest.mark.parametrize("N,angle", [(64, True), (128, True), (96, False)])
def test_integrate_identity_pose(ctx, oracle_mod, small_scene, N, angle):
    cam, frames, _ = small_scene
    depth, rgb = frames[0]
    _, _, nmap = _maps(oracle_mod, cam, depth)
    size = 6.0
    trunc = max(0.06, 2.1 * size / N)
    (vol, col, scaled, U), (gvol, gcol, gscaled), _ = _integrate_both(ctx, oracle_mod, cam, depth, rgb, nmap, N, size, np.eye(3), [3, 3, 3], trunc,
                                                                    [0, 0, 0], angle)
    assert U > 1000
    assert np.array_equal(scaled.view(np.uint32), gscaled.view(np.uint32))
    assert np.array_equal(vol, gvol)
    assert np.array_equal(col, gcol)


def test_integrate_random_poses_wrapped_and_accumulated(ctx, oracle_mod, small_scene):
    """Several frames into the same volume with rotated poses and a non-zero storage wrap (non power-of-two N)."""
    cam, frames, _ = small_scene
    rng = np.random.default_rng(7)
    N, size = 80, 6.0
    trunc = max(0.06, 2.1 * size / N)
    vol = np.zeros((N, N, N), np.int16)
    col = np.zeros((N, N, N, 4), np.uint8)
    for k in range(4):
        depth, rgb = frames[k]
        _, _, nmap = _maps(oracle_mod, cam, depth)
        R = random_rotation(rng, 0.5)
        Rinv = oracle_mod.mat33_inverse(R)
        t = (np.array([3, 3, 3]) + rng.uniform(-0.4, 0.4, 3)).astype(np.float32)
        wrap = [int(w) for w in rng.integers(0, N, 3)]
        (vol2, col2, _, U), (gvol, gcol, _), _ = _integrate_both(ctx, oracle_mod, cam, depth, rgb, nmap, N, size, Rinv, t, trunc, wrap, True, vol, col)
        assert U > 100
        assert np.array_equal(vol2, gvol), f"frame {k}: {(vol2 != gvol).sum()} tsdf mismatches"
        assert np.array_equal(col2, gcol), f"frame {k}: {(col2 != gcol).sum()} colour mismatches"
        vol, col = vol2, col2


def test_integrate_camera_outside_and_empty(ctx, oracle_mod, small_scene):
    cam, frames, _ = small_scene
    depth, rgb = frames[0]
    _, _, nmap = _maps(oracle_mod, cam, depth)
    N, size = 64, 6.0
    # static-mode style pose: camera 0.45 m outside the near face
    (vol, col, _, U), (gvol, gcol, _), _ = _integrate_both(ctx, oracle_mod, cam, depth, rgb, nmap, N, size, np.eye(3), [3, 3, -0.45], 0.2, [0, 0, 0], True)
    assert U > 100 and np.array_equal(vol, gvol) and np.array_equal(col, gcol)
    # all-zero depth: nothing may change
    z = np.zeros_like(depth)
    (vol, col, _, U), (gvol, gcol, _), _ = _integrate_both(ctx, oracle_mod, cam, z, rgb, nmap, N, size, np.eye(3), [3, 3, 3], 0.2, [0, 0, 0], True)
    assert U == 0 and not gvol.any() and not gcol.any()


def _fused_volume(oracle, cam, frames, traj, N, size, trunc, nframes=3):
    from oracle.oracle import OIntr
    vol = np.zeros((N, N, N), np.int16)
    col = np.zeros((N, N, N, 4), np.uint8)
    for k in range(nframes):
        depth, rgb = frames[k]
        _, _, nmap = _maps(oracle, cam, depth)
        R, c = traj[k]
        Rf = R.astype(np.float32)
        oracle.integrate_tsdf(depth, OIntr(cam.fx, cam.fy, cam.cx, cam.cy), [size] * 3, oracle.mat33_inverse(Rf), (c + size / 2).astype(np.float32),
                              trunc, vol, [0, 0, 0], col, rgb, nmap, True)
    return vol, col


def _rotate_storage(vol, wrap):
    """logical volume -> storage layout for a given wrap (storage[(i + w) % N] = logical[i])"""
    return np.roll(vol, shift=(wrap[2], wrap[1], wrap[0]), axis=(0, 1, 2))


@pytest.mark.parametrize("wrap", [[0, 0, 0], [17, 5, 40]])
def test_raycast(ctx, oracle_mod, small_scene, wrap):
    from kintinuous_amd.abi import Intr
    from oracle.oracle import OIntr
    cam, frames, traj = small_scene
    N, size = 96, 6.0
    trunc = max(0.06, 2.1 * size / N)
    vol, col = _fused_volume(oracle_mod, cam, frames, traj, N, size, trunc)
    vol, col = _rotate_storage(vol, wrap), _rotate_storage(col, wrap)
    R, c = traj[2]
    Rf, t = R.astype(np.float32), (c + size / 2).astype(np.float32)
    rows, cols = cam.rows, cam.cols
    rng = np.random.default_rng(3)
    # pre-fill outputs with a pattern: planes the kernel must NOT touch (y/z of unhit pixels) have to survive
    vmap0 = rng.uniform(-1, 1, (3 * rows, cols)).astype(np.float32)
    nmap0 = rng.uniform(-1, 1, (3 * rows, cols)).astype(np.float32)
    colr0 = rng.integers(0, 255, (rows, cols, 4)).astype(np.uint8)
    vmap, nmap, colr = vmap0.copy(), nmap0.copy(), colr0.copy()
    S = oracle_mod.raycast(OIntr(cam.fx, cam.fy, cam.cx, cam.cy), Rf, t, trunc, [size] * 3, vol, vmap, nmap, wrap, colr, col)
    dv, dn, dc = ctx.upload(vmap0), ctx.upload(nmap0), ctx.upload(colr0)
    ctx.raycast(Intr(cam.fx, cam.fy, cam.cx, cam.cy), Rf, t, trunc, [size] * 3, ctx.upload(vol), dv, dn, cols, rows, wrap, dc, ctx.upload(col), N)
    ctx.sync()
    gv, gn, gc = ctx.download(dv, np.float32, vmap.shape), ctx.download(dn, np.float32, nmap.shape), ctx.download(dc, np.uint8, colr.shape)
    assert S > rows * cols
    hits = np.isfinite(vmap[:rows]).sum()
    assert hits > 0.5 * rows * cols
    assert np.array_equal(vmap.view(np.uint32), gv.view(np.uint32))
    assert np.array_equal(nmap.view(np.uint32), gn.view(np.uint32))
    assert np.array_equal(colr, gc)


def test_raycast_camera_outside_volume(ctx, oracle_mod, small_scene):
    from kintinuous_amd.abi import Intr
    from oracle.oracle import OIntr
    cam, frames, traj = small_scene
    N, size = 64, 6.0
    depth, rgb = frames[0]
    _, _, nmap = _maps(oracle_mod, cam, depth)
    vol = np.zeros((N, N, N), np.int16)
    col = np.zeros((N, N, N, 4), np.uint8)
    t = np.array([3, 3, -0.45], np.float32)
    oracle_mod.integrate_tsdf(depth, OIntr(cam.fx, cam.fy, cam.cx, cam.cy), [size] * 3, np.eye(3), t, 0.2, vol, [0, 0, 0], col, rgb, nmap, True)
    rows, cols = cam.rows, cam.cols
    vmap, nmap_o, colr = np.zeros((3 * rows, cols), np.float32), np.zeros((3 * rows, cols), np.float32), np.zeros((rows, cols, 4), np.uint8)
    oracle_mod.raycast(OIntr(cam.fx, cam.fy, cam.cx, cam.cy), np.eye(3), t, 0.2, [size] * 3, vol, vmap, nmap_o, [0, 0, 0], colr, col)
    dv, dn, dc = ctx.zeros(vmap.nbytes), ctx.zeros(vmap.nbytes), ctx.zeros(colr.nbytes)
    ctx.raycast(Intr(cam.fx, cam.fy, cam.cx, cam.cy), np.eye(3), t, 0.2, [size] * 3, ctx.upload(vol), dv, dn, cols, rows, [0, 0, 0], dc, ctx.upload(col), N)
    ctx.sync()
    assert np.array_equal(vmap.view(np.uint32), ctx.download(dv, np.float32, vmap.shape).view(np.uint32))
    assert np.array_equal(nmap_o.view(np.uint32), ctx.download(dn, np.float32, vmap.shape).view(np.uint32))
    assert np.array_equal(colr, ctx.download(dc, np.uint8, colr.shape))


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("back", [False, True])
@pytest.mark.parametrize("cur,delta", [(0, 14), (3, 14), (-5, 14), (60, 14), (0, 16), (-30, 3), (50, 15)])
def test_clear_volume(ctx, oracle_mod, axis, back, cur, delta):
    N = 64
    rng = np.random.default_rng(axis * 7 + abs(cur))
    d = -delta if back else delta
    for dtype in (np.int16, np.uint32):
        vol = rng.integers(1, 30000, (N, N, N)).astype(dtype)
        ref = vol.copy()
        oracle_mod.clear_volume(ref if dtype == np.int16 else ref.view(np.uint32), axis, back, cur, cur + d)
        dv = ctx.upload(vol)
        ctx.clear_volume(dv, vol.dtype.itemsize, N, axis, back, cur, cur + d)
        ctx.sync()
        got = ctx.download(dv, dtype, vol.shape)
        assert (ref == 0).sum() > 0
        assert np.array_equal(ref, got), f"cleared {int((ref == 0).sum())} vs {int((got == 0).sum())}"


def _sorted_points(p):
    k = np.stack([p["xyz"][:, 0], p["xyz"][:, 1], p["xyz"][:, 2]], 1).view(np.uint32).astype(np.uint64)
    key = np.lexsort((p["bgra"].view(np.uint32)[:, 0], k[:, 2], k[:, 1], k[:, 0]))
    return p[key]


@pytest.mark.parametrize("box", ["xplus", "xminus", "yplus", "zminus", "full", "sub2"])
def test_extract_cloud_slice(ctx, oracle_mod, small_scene, box):
    cam, frames, traj = small_scene
    N, size = 96, 6.0
    trunc = max(0.06, 2.1 * size / N)
    vol, col = _fused_volume(oracle_mod, cam, frames, traj, N, size, trunc)
    wrap = [11, 0, 90]
    vol, col = _rotate_storage(vol, wrap), _rotate_storage(col, wrap)
    real = [11 - 2 * N, 0, 90]  # realVoxelWrap may be negative / beyond N; the storage wrap is its normalised copy
    lo, hi, sub = [0, 0, 0], [N, N, N], 1
    if box == "xplus": hi[0] = 14 + 1 + 2
    elif box == "xminus": lo[0] = N + (-14 - 2)
    elif box == "yplus": hi[1] = 17
    elif box == "zminus": lo[2], hi[2] = N + (-14 - 2) - 1, N - 1
    elif box == "sub2": sub = 2
    cap = 400000
    ref = oracle_mod.extract_cloud_slice(vol, [size] * 3, cap, wrap, col, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], sub, real)
    out = ctx.empty(cap * 32)
    n = ctx.extract_cloud_slice(ctx.upload(vol), [size] * 3, out, cap, wrap, ctx.upload(col), lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], sub, real, N)
    from kintinuous_amd.abi import POINT_DTYPE
    got = ctx.download(out, POINT_DTYPE, (cap,))[:n]
    assert n == len(ref)
    if box == "full":
        assert n > 1000
    a, b = _sorted_points(ref), _sorted_points(got)
    assert np.array_equal(a["xyz"].view(np.uint32), b["xyz"].view(np.uint32))
    assert np.array_equal(a["bgra"], b["bgra"])


def test_extract_capacity_clamp(ctx, oracle_mod, small_scene):
    cam, frames, traj = small_scene
    N, size = 64, 6.0
    vol, col = _fused_volume(oracle_mod, cam, frames, traj, N, size, 0.2, nframes=1)
    cap = 100
    out = ctx.empty(cap * 32)
    n = ctx.extract_cloud_slice(ctx.upload(vol), [size] * 3, out, cap, [0, 0, 0], ctx.upload(col), 0, N, 0, N, 0, N, 1, [0, 0, 0], N)
    assert n == cap


def test_unpack_tsdf_all_shorts(ctx):
    """The device restates (float)v / 32767 as a multiply plus two FMAs: must equal the IEEE division for all 65536 shorts."""
    import ctypes as C
    from kintinuous_amd import abi
    out = np.zeros(65536, np.float32)
    abi._chk(abi.lib().kt_debug_unpack_table(ctx.h, out.ctypes.data_as(C.POINTER(C.c_float))))
    v = np.arange(-32768, 32768, dtype=np.float32)
    ref = (v / np.float32(32767)).astype(np.float32)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_exact_reciprocal_all_floats(ctx):
    """tsdf23's 1 / d is the IEEE refinement chain without its scaling wrapper (kt_rcp_exact): equal to the division for every float
    in the range the kernel uses it on, 2^-20 <= |d| <= 2^20 (2 x 3.4e8 values)."""
    import ctypes as C
    from kintinuous_amd import abi
    bad = C.c_uint(12345)
    abi._chk(abi.lib().kt_debug_rcp_check(ctx.h, C.byref(bad)))
    assert bad.value == 0


def test_running_average_division_all_floats(ctx):
    """tsdf23 forms (F W + tsdf) / (W + 1) as q = n y, q' = fma(fma(-d, q, n), y, q) with y = RN(1 / d) from a table (Markstein's
    correction).  Compared on the device with the IEEE division for EVERY finite float numerator and every divisor 1..256 (1.1e12
    pairs): identical wherever |n| >= 2^-100; below that the two may differ in the last bit of a quotient of magnitude < 2^-100, which
    pack_tsdf (x 32767, truncate) maps to 0 either way."""
    import ctypes as C
    from kintinuous_amd import abi
    out = (C.c_uint * 3)(1, 1, 1)
    abi._chk(abi.measure_lib().kt_debug_div_check(ctx.h, out))
    assert out[2] == 0, list(out)
    assert out[1] < 0x0d800000, hex(out[1])   # every differing numerator is below 2^-100
