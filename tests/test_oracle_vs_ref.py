"""Pins the oracle to the reference: every kernel of oracle/kt_oracle_kernels.c against oracle/_ref/libkt_ref.so, which is
the reference's OWN source files (frontend/cuda/*.cu + containers/device_memory.cpp) compiled for the CPU by oracle/Makefile
against the CUDA emulation in oracle/ref_shim/ (fibers for threads, rendez-vous for __syncthreads / __shfl_down / __ballot).

Bar: bit-exact -- integers, bytes, indices AND floats (the oracle's hand-placed fmaf() sites must be the contractions
clang -ffp-contract=fast makes on the reference source; both sides use IEEE division / sqrt, rsqrtf = 1/sqrtf, denormals
flushed inside kernels as --ftz=true does).  The two documented exceptions are asserted as such, not tolerated silently:
  * __expf: _ref uses libm expf, the oracle its own kto_expf -- the bilateral filter output is identical except where the
    last bit of a weight flips a tie of rn(sum1/sum2) (bounded: <= 2e-5 of the pixels, 1 mm);
  * (unsigned char)NaN is undefined in C++ (CUDA gives 0): the raycast "heat" byte of pixels whose vertex lies within one
    voxel of a volume face is excluded.
What _ref cannot pin: nvcc's --prec-div=false / --prec-sqrt=false / ex2.approx (hardware approximations) and the
host-side code of the path (Eigen / OpenCV), see DESIGN.md section 5.
"""
import os

import numpy as np
import pytest

from conftest import random_rotation, random_volume_state

pytestmark = pytest.mark.skipif(not __import__("oracle.ref", fromlist=["available"]).available(), reason="oracle/_ref not built and /root/reference absent")


@pytest.fixture(scope="module")
def R():
    from oracle import ref
    ref.build()
    ref.lib()
    return ref


def same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    return np.array_equal(a.view(np.uint8), b.view(np.uint8))


def nmism(a, b):
    return int((np.ascontiguousarray(a).view(np.uint8) != np.ascontiguousarray(b).view(np.uint8)).sum())


def _frames(cols, rows, n, scene="room"):
    from kintinuous_amd import synth
    cam = synth.Camera.small(cols, rows) if (cols, rows) != (640, 480) else synth.Camera()
    sc = synth.Scene(scene)
    traj = synth.orbit_trajectory(n)
    return cam, [synth.render(sc, cam, Rm, c) for (Rm, c) in traj], traj


def _holes(depth, rng, frac=0.03):
    d = depth.copy()
    d[rng.random(d.shape) < frac] = 0
    return d


@pytest.mark.parametrize("cols,rows", [(160, 120), (96, 70), (640, 480)])
def test_image_kernels(oracle_mod, R, cols, rows):
    """a1-a5, a13 + the RGB-D pyramids (bilateral_pyrdown.cu, maps.cu) on ragged, VGA and hole-ridden inputs."""
    O = oracle_mod
    from oracle.oracle import OIntr
    rng = np.random.default_rng(cols)
    cam, frames, _ = _frames(cols, rows, 1)
    depth, rgb = frames[0]
    depth = _holes(depth, rng)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    bo = O.bilateral_filter(depth)
    br = R.bilateral_filter(depth)
    # __expf model (libm expf in _ref, kto_expf in the oracle, ex2.approx in CUDA): the last bit of a weight can flip a tie of
    # rn(sum1 / sum2) -- at most 2e-5 of the pixels, by 1 mm
    bad = bo != br
    assert int(bad.sum()) <= max(1, bo.size // 50000) and (np.abs(bo.astype(int) - br.astype(int)) <= 1).all()
    noisy = rng.integers(0, 65536, depth.shape).astype(np.uint16)      # full u16 range: the int product (value - tmp)^2 wraps
    if cols < 640:
        assert same(O.bilateral_filter(noisy), R.bilateral_filter(noisy))
        # white noise makes every off-centre weight tiny, so the last bit of exp() decides rn(sum1 / sum2) now and then: this
        # is the __expf model (libm expf in _ref, kto_expf in the oracle, ex2.approx in CUDA), bounded here, exact on scenes
        noisy2 = rng.integers(0, 40000, depth.shape).astype(np.uint16)
        assert int((O.bilateral_filter(noisy2) != R.bilateral_filter(noisy2)).sum()) <= max(2, noisy2.size // 2000)
    assert same(O.pyr_down(bo), R.pyr_down(bo))
    assert same(O.pyr_down(noisy), R.pyr_down(noisy))
    vo = O.create_vmap(intr, bo)
    assert same(vo, R.create_vmap(intr, bo))
    no = O.create_nmap(vo)
    assert same(no, R.create_nmap(vo))
    Rm = O.rodrigues(np.array([0.1, 0.2, -0.05])).astype(np.float32)
    a, b = O.transform_maps(vo, no, Rm, [0.1, -0.2, 0.3])
    c, d = R.transform_maps(vo, no, Rm, [0.1, -0.2, 0.3])
    assert same(a, c) and same(b, d)
    assert same(O.resize_map(vo, False), R.resize_map(vo, False))
    assert same(O.resize_map(no, True), R.resize_map(no, True))
    dm = O.depth_to_metres(depth, 6000)
    assert same(dm, R.depth_to_metres(depth, 6000))
    io = O.bgr_to_intensity(rgb)
    assert same(io, R.bgr_to_intensity(rgb))
    assert same(O.pyr_down_gauss_f32(dm), R.pyr_down_gauss_f32(dm))
    assert same(O.pyr_down_gauss_u8(io), R.pyr_down_gauss_u8(io))
    rnd8 = rng.integers(0, 256, io.shape).astype(np.uint8)
    assert same(O.pyr_down_gauss_u8(rnd8), R.pyr_down_gauss_u8(rnd8))
    for img in (io, rnd8):
        dxo, dyo = O.derivative_images(img)
        dxr, dyr = R.derivative_images(img)
        assert same(dxo, dxr) and same(dyo, dyr)
    for level in (0, 2):
        assert same(O.project_to_cloud(dm, cam.fx, cam.fy, cam.cx, cam.cy, level), R.project_to_cloud(dm, cam.fx, cam.fy, cam.cx, cam.cy, level))


def test_generate_image_and_depth(oracle_mod, R):
    """image_generator.cu (live-view products)."""
    O = oracle_mod
    from oracle.oracle import OIntr
    cam, frames, _ = _frames(160, 120, 1)
    depth, rgb = frames[0]
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    v = O.create_vmap(intr, O.bilateral_filter(depth))
    n = O.create_nmap(v)
    col = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 7, np.uint8)], axis=2)
    a, b = O.generate_image(v, n, col, [0.5, -1.0, -2.0])
    c, d = R.generate_image(v, n, col, [0.5, -1.0, -2.0])
    assert same(a, c) and same(b, d)
    Rinv = O.mat33_inverse(O.rodrigues(np.array([0.02, -0.03, 0.01])).astype(np.float32))
    assert same(O.generate_depth(Rinv, [0.1, 0.0, -0.2], v, n), R.generate_depth(Rinv, [0.1, 0.0, -0.2], v, n, 6.0))


def _volume_after(O, cam, frames, traj, N, size, nframes, wrap, angle=True, rng=None):
    from oracle.oracle import OIntr
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    trunc = max(0.06 if size == 6.0 else max(0.01, size / 100), 2.1 * size / N)
    vol = np.zeros((N, N, N), np.int16)
    col = np.zeros((N, N, N, 4), np.uint8)
    poses = []
    for k in range(nframes):
        d, c = frames[k]
        v = O.create_vmap(intr, O.bilateral_filter(d))
        n = O.create_nmap(v)
        Rk = np.asarray(traj[k][0], np.float32)
        tk = (np.asarray(traj[k][1], np.float32) + np.float32(size / 2)).astype(np.float32)
        if rng is not None:
            Rk = (random_rotation(rng, 0.3) @ Rk).astype(np.float32)
            tk = (tk + rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
        O.integrate_tsdf(d, intr, [size] * 3, O.mat33_inverse(Rk), tk, trunc, vol, wrap, col, c, n, angle)
        poses.append((Rk, tk))
    return intr, trunc, vol, col, poses


@pytest.mark.parametrize("N,wrap,angle", [(64, [0, 0, 0], True), (96, [5, 90, 41], True), (80, [79, 1, 33], False)])
def test_integrate(oracle_mod, R, N, wrap, angle):
    """a11: scaleDepth + tsdf23 over several frames into the same volume (accumulated weights and colours), random poses,
    wrapped storage, non-power-of-two N."""
    O = oracle_mod
    from oracle.oracle import OIntr
    rng = np.random.default_rng(N)
    cam, frames, traj = _frames(160, 120, 4)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    size = 6.0
    trunc = max(0.06, 2.1 * size / N)
    vo, co = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    vr, cr = vo.copy(), co.copy()
    for k in range(4):
        d, c = frames[k]
        d = _holes(d, rng, 0.01)
        v = O.create_vmap(intr, O.bilateral_filter(d))
        n = O.create_nmap(v)
        Rk = (random_rotation(rng, 0.4) @ np.asarray(traj[k][0], np.float32)).astype(np.float32) if k else np.eye(3, dtype=np.float32)
        tk = (np.array([3, 3, 3]) + (rng.uniform(-0.4, 0.4, 3) if k else 0)).astype(np.float32)
        Rinv = O.mat33_inverse(Rk)
        U, so = O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, angle)
        sr = R.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vr, wrap, cr, c, n, angle)
        assert U > 100
        assert same(so, sr), f"frame {k}: scaleDepth differs at {nmism(so, sr)} bytes"
        assert same(vo, vr), f"frame {k}: {int((vo != vr).sum())} tsdf mismatches"
        assert same(co, cr), f"frame {k}: {int((co != cr).any(axis=-1).sum())} colour / weight mismatches"


def test_integrate_camera_outside(oracle_mod, R):
    O = oracle_mod
    from oracle.oracle import OIntr
    cam, frames, _ = _frames(160, 120, 1)
    d, c = frames[0]
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
    N = 64
    vo, co = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    vr, cr = vo.copy(), co.copy()
    U, _ = O.integrate_tsdf(d, intr, [6.0] * 3, np.eye(3), [3, 3, -0.45], 0.2, vo, [0, 0, 0], co, c, n, True)   # static mode pose
    R.integrate_tsdf(d, intr, [6.0] * 3, np.eye(3), [3, 3, -0.45], 0.2, vr, [0, 0, 0], cr, c, n, True)
    assert U > 100 and same(vo, vr) and same(co, cr)


def test_init_volumes(oracle_mod, R):
    O = oracle_mod
    N = 32
    vo, co = np.full((N, N, N), 0x5A5A, np.int16), np.full((N, N, N, 4), 0x5A, np.uint8)
    O.lib().kto_init_volume(vo.ctypes.data_as(__import__('ctypes').c_void_p), N)
    O.lib().kto_init_color_volume(co.ctypes.data_as(__import__('ctypes').c_void_p), N)
    assert same(vo, R.init_volume(N)) and same(co, R.init_color_volume(N))


@pytest.mark.parametrize("N,wrap", [(64, [0, 0, 0]), (96, [5, 90, 41])])
def test_raycast(oracle_mod, R, N, wrap):
    """a12: vertex / normal maps bit-exact, colour bytes exact; the heat byte is excluded where the reference converts NaN
    to unsigned char (undefined behaviour on the CPU, 0 in CUDA): vertices within one voxel of a face."""
    O = oracle_mod
    rng = np.random.default_rng(N + 1)
    cam, frames, traj = _frames(160, 120, 4)
    size = 6.0
    intr, trunc, vol, col, poses = _volume_after(O, cam, frames, traj, N, size, 3, wrap)
    cell = np.float32(size / N)
    for trial in range(3):
        Rk, tk = poses[trial]
        if trial == 2:
            Rk = (random_rotation(rng, 0.2) @ Rk).astype(np.float32)
            tk = (tk + rng.uniform(-0.2, 0.2, 3)).astype(np.float32)
        outs = []
        for M in (O, R):
            vm = np.full((3 * cam.rows, cam.cols), 7.0, np.float32)   # pre-filled: untouched planes must stay untouched
            nm = np.full_like(vm, -3.0)
            cm = np.full((cam.rows, cam.cols, 4), 9, np.uint8)
            M.raycast(intr, Rk, tk, trunc, [size] * 3, vol, vm, nm, wrap, cm, col)
            outs.append((vm, nm, cm))
        (vo, no, co), (vr, nr, cr) = outs
        assert np.isfinite(vo[: cam.rows]).sum() > 1000
        assert same(vo, vr), f"vmap: {nmism(vo, vr)} bytes"
        assert same(no, nr), f"nmap: {nmism(no, nr)} bytes"
        assert same(co[..., :3], cr[..., :3])
        hit = np.isfinite(vo[: cam.rows])
        g = np.floor(np.stack([vo[: cam.rows], vo[cam.rows: 2 * cam.rows], vo[2 * cam.rows:]], -1) / cell)
        border = hit & ((g <= 0) | (g >= N - 1)).any(axis=-1)
        assert same(co[..., 3][~border], cr[..., 3][~border])
        assert border.sum() < 0.02 * hit.sum()


def test_raycast_camera_outside(oracle_mod, R):
    O = oracle_mod
    from oracle.oracle import OIntr
    cam, frames, _ = _frames(160, 120, 1)
    d, c = frames[0]
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
    N = 64
    vol, col = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    O.integrate_tsdf(d, intr, [6.0] * 3, np.eye(3), [3, 3, -0.45], 0.2, vol, [0, 0, 0], col, c, n, True)
    outs = []
    for M in (O, R):
        vm = np.zeros((3 * cam.rows, cam.cols), np.float32)
        nm = np.zeros_like(vm)
        cm = np.zeros((cam.rows, cam.cols, 4), np.uint8)
        M.raycast(intr, np.eye(3), [3, 3, -0.45], 0.2, [6.0] * 3, vol, vm, nm, [0, 0, 0], cm, col)
        outs.append((vm, nm, cm))
    assert np.isfinite(outs[0][0][: cam.rows]).sum() > 1000
    assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1]) and same(outs[0][2][..., :3], outs[1][2][..., :3])


@pytest.mark.parametrize("elem", [np.int16, np.uint32])
@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("back", [False, True])
def test_clear_volume(oracle_mod, R, elem, axis, back):
    """a14 incl. the X launch-geometry quirk (delta 16 -> 17-plane slab, one plane left) and wraps across the seam."""
    O = oracle_mod
    N = 48
    rng = np.random.default_rng(axis * 2 + back)
    cases = [(0, 14), (40, 54), (-3, 11), (100, 116), (47, 61), (-20, -6)] if not back else [(0, -14), (10, -4), (-40, -54), (100, 84), (3, -13)]
    for cur, delta in cases:
        a = rng.integers(1, 100, (N, N, N)).astype(elem)
        if elem == np.uint32:
            a = a.view(np.uint8).reshape(N, N, N, 4).copy()
        b = a.copy()
        O.clear_volume(a, axis, back, cur, delta)
        R.clear_volume(b, axis, back, cur, delta)
        assert same(a, b), (axis, back, cur, delta, nmism(a, b))
        assert (a == 0).any()


def _point_set(p):
    """slices are compared as sets of (xyz bits, bgra): the reference's output order depends on atomicAdd timing"""
    raw = np.ascontiguousarray(p).view(np.uint8).reshape(len(p), 32)
    key = np.concatenate([raw[:, :12], raw[:, 16:20]], axis=1)
    return sorted(map(bytes, key))


@pytest.mark.parametrize("box", ["xplus", "xminus", "yplus", "zminus", "full", "sub2"])
def test_extract_cloud_slice(oracle_mod, R, box):
    """a15: warp-compacted zero-crossing extraction (ballot / popc / scan_warp / atomicAdd in the reference)."""
    O = oracle_mod
    N, size = 64, 6.0
    wrap = [7, 60, 13]
    cam, frames, traj = _frames(160, 120, 3)
    _, _, vol, col, _ = _volume_after(O, cam, frames, traj, N, size, 3, wrap)
    real = [3, -2, 70]
    sub = 1
    if box == "xplus":
        b = (0, 17, 0, N, 0, N)
    elif box == "xminus":
        b = (N - 12, N, 0, N, 0, N)
    elif box == "yplus":
        b = (0, N, 0, 17, 0, N)
    elif box == "zminus":
        b = (0, N, 0, N, N - 13, N)      # KintinuousTracker.cpp:805 off-by-one slab
    elif box == "full":
        b = (0, N, 0, N, 0, N)
    else:
        b, sub = (0, N, 0, N, 0, N), 2
    po = O.extract_cloud_slice(vol, [size] * 3, 400000, wrap, col, *b, sub, real)
    pr = R.extract_cloud_slice(vol, [size] * 3, 400000, wrap, col, *b, sub, real)
    assert len(po) == len(pr)
    if box in ("full", "zminus"):
        assert len(po) > 500
    assert _point_set(po) == _point_set(pr)


@pytest.mark.parametrize("cols,rows", [(160, 120), (640, 480)])
def test_icp_step(oracle_mod, R, cols, rows):
    """a6: correspondence search + 29-float reduction through the reference's own icpKernel<<<64,128>>> + reduceSum<<<1,512>>>
    (the __shfl_down trees run on the fiber emulation): A, b, residual bit-identical to the oracle's reference-order mode."""
    O = oracle_mod
    N, size = 96, 6.0
    cam, frames, traj = _frames(cols, rows, 4)
    intr, trunc, vol, col, poses = _volume_after(O, cam, frames, traj, N, size, 2, [0, 0, 0])
    Rp, tp = poses[1]
    vprev = np.zeros((3 * rows, cols), np.float32)
    nprev = np.zeros_like(vprev)
    O.raycast(intr, Rp, tp, trunc, [size] * 3, vol, vprev, nprev, [0, 0, 0], np.zeros((rows, cols, 4), np.uint8), col)
    d, _ = frames[3]
    v = O.create_vmap(intr, O.bilateral_filter(d))
    n = O.create_nmap(v)
    Rp_inv = O.mat33_inverse(Rp)
    rng = np.random.default_rng(5)
    for trial in range(2):
        Rc = Rp if trial == 0 else (random_rotation(rng, 0.02) @ Rp).astype(np.float32)
        tc = tp if trial == 0 else (tp + rng.uniform(-0.01, 0.01, 3)).astype(np.float32)
        Ao, bo, ro = O.icp_step(Rc, tc, v, n, Rp_inv, tp, intr, vprev, nprev, 0.10, float(np.sin(np.float32(20.0 * 3.14159265 / 180.0))), 0)
        Ar, br, rr = R.icp_step(Rc, tc, v, n, Rp_inv, tp, intr, vprev, nprev, 0.10, float(np.sin(np.float32(20.0 * 3.14159265 / 180.0))))
        assert ro[1] > 0.3 * cols * rows
        assert same(Ao, Ar) and same(bo, br) and same(ro, rr), (Ao - Ar, bo - br, ro, rr)


@pytest.mark.parametrize("level", [0, 1])
def test_rgb_residual_and_step(oracle_mod, R, level):
    """a8, a9: residualKernel<<<256,128>>> + int2 reduceSum, rgbKernel + reduceSum: DataTerm of every valid pixel, count, sigma,
    A, b bit-identical."""
    O = oracle_mod
    cols, rows = 160 >> level, 120 >> level
    cam, frames, _ = _frames(160, 120, 8)
    pyr = []
    for d, rgb in (frames[0], frames[1]):
        dm, it = O.depth_to_metres(d, 6000), O.bgr_to_intensity(rgb)
        for _ in range(level):
            dm, it = O.pyr_down_gauss_f32(dm), O.pyr_down_gauss_u8(it)
        pyr.append((dm, it))
    (ld, li), (nd, ni) = pyr
    dx, dy = O.derivative_images(ni)
    f = 1.0 / (1 << level)
    fx, fy, cx, cy = cam.fx * f, cam.fy * f, cam.cx * f, cam.cy * f
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    Rm = O.rodrigues(np.array([0.002, -0.003, 0.001]))
    krkinv = (K @ Rm @ np.linalg.inv(K)).astype(np.float32)
    kt = (K @ np.array([0.004, -0.002, 0.003])).astype(np.float32)
    min_scale = (np.float32([12, 5, 3, 1][level]) / np.float32(0.125)) ** 2
    co, so, no = O.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, 0.07, kt, krkinv)
    cr, sr, nr = R.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, 0.07, kt, krkinv)
    assert no > 50 and (so, no) == (sr, nr)
    assert np.array_equal(co["valid"], cr["valid"] != 0)
    m = co["valid"] != 0
    for f_ in ("zero", "one", "diff"):
        assert same(co[f_][m], cr[f_][m])
    cloud = O.project_to_cloud(ld, fx, fy, cx, cy, 0)
    sigma = float(np.sqrt(np.float32(no)))   # RGBDOdometry.cpp:253 quirk
    Ao, bo = O.rgb_step(co, sigma, cloud, fx, fy, dx, dy, 0.125, 0)
    Ar, br = R.rgb_step(cr, sigma, cloud, fx, fy, dx, dy, 0.125)
    assert same(Ao, Ar) and same(bo, br), (Ao - Ar, bo - br)


def _extract_geometry_ok(N, lo, hi):
    """extractCloudSlice's launch (extract.cu:357-412) offsets its grid to the slab when the box is thinnest in x or in y and rounds the
    slab up to 16; a block that then lies entirely outside the volume returns before it has counted itself in `blocks_done`, the
    last-block bookkeeping never runs, the call returns the PREVIOUS call's count and leaves its own in `global_count` for the next
    one.  The tracker's slabs never do that (N = 512, slabs of 16 + overlap); a random box may.  True when every block has a row / column
    inside the volume."""
    ax, ay, az = hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]
    if ax == N and ay == N and az == N:
        return True
    r16 = lambda v: v if v % 16 == 0 else v + 16 - v % 16
    if ax < ay and ax < az:
        gx = -(-r16(ax) // 32)
        return lo[0] + 32 * (gx - 1) < N
    if ay < ax and ay < az:
        gy = -(-r16(ay) // 6)
        return lo[1] + 6 * (gy - 1) < N
    return True


@pytest.mark.parametrize("seed", list(range(16)))
def test_randomized_sweep(oracle_mod, R, seed):
    """The same comparison over seeded random configurations -- scene, image size (including sizes that are no multiple of the tile and
    block shapes), volume resolution and extent, storage wrap, poses with large rotations, holes and sensor noise in the depth, the
    colour-angle flag -- through integrate (4 frames into one volume), raycast (2 poses), extraction (a random box), a slab clear and
    one ICP reduction: every output bit for bit, the one documented exception aside (the heat byte next to a volume face)."""
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O = oracle_mod
    rng = np.random.default_rng(1000 + seed)
    cols, rows = [(160, 120), (200, 150), (136, 104), (320, 240)][seed % 4]
    N = int(rng.choice([48, 64, 72, 100]))
    size = float(rng.choice([4.0, 6.0, 7.0]))
    wrap = [int(v) for v in rng.integers(0, N, 3)]
    angle = bool(rng.integers(0, 2))
    kind = ["room", "wall", "room", "farwall"][seed % 4]
    cam = synth.Camera.small(cols, rows)
    scene = synth.Scene(kind, seed=1234 + seed)
    base = synth.orbit_trajectory(40)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    trunc = max(0.06 if size == 6.0 else max(0.01, size / 100), 2.1 * size / N)
    vo, co = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    vr, cr = vo.copy(), co.copy()
    poses, maps = [], []
    for k in range(4):
        Rm, c0 = base[int(rng.integers(0, 40))]
        d, c = synth.render(scene, cam, Rm, c0, noise_mm=float(rng.choice([0.0, 1.5])), rng=rng)
        d = _holes(d, rng, float(rng.choice([0.0, 0.02])))
        Rk = (random_rotation(rng, 0.5) @ np.asarray(Rm, np.float32)).astype(np.float32)
        tk = (np.asarray(c0, np.float32) + np.float32(size / 2) + rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
        v = O.create_vmap(intr, O.bilateral_filter(d))
        n = O.create_nmap(v)
        Rinv = O.mat33_inverse(Rk)
        U, so = O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, angle)
        sr = R.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vr, wrap, cr, c, n, angle)
        assert same(so, sr) and same(vo, vr) and same(co, cr), (seed, k, int((vo != vr).sum()), int((co != cr).any(axis=-1).sum()))
        poses.append((Rk, tk))
        maps.append((v, n))
    assert int((co[..., 3] != 0).sum()) > 200
    # raycast from two of the poses
    cell = np.float32(size / N)
    pred = None
    for Rk, tk in poses[2:]:
        outs = []
        for M in (O, R):
            vm, nm = np.full((3 * rows, cols), 7.0, np.float32), np.full((3 * rows, cols), -3.0, np.float32)
            cm = np.full((rows, cols, 4), 9, np.uint8)
            M.raycast(intr, Rk, tk, trunc, [size] * 3, vo, vm, nm, wrap, cm, co)
            outs.append((vm, nm, cm))
        (a, b, c_), (a2, b2, c2) = outs
        assert same(a, a2) and same(b, b2) and same(c_[..., :3], c2[..., :3]), seed
        hit = np.isfinite(a[:rows])
        g = np.floor(np.stack([a[:rows], a[rows: 2 * rows], a[2 * rows:]], -1) / cell)
        border = hit & ((g <= 0) | (g >= N - 1)).any(axis=-1)
        assert same(c_[..., 3][~border], c2[..., 3][~border])
        pred = (Rk, tk, a, b)
    # one ICP reduction of the last frame's maps against the last prediction, from a perturbed pose
    Rp, tp, vprev, nprev = pred
    Rc = (random_rotation(rng, 0.03) @ Rp).astype(np.float32)
    tc = (tp + rng.uniform(-0.02, 0.02, 3)).astype(np.float32)
    th = float(np.sin(np.float32(20.0 * 3.14159265 / 180.0)))
    Ao, bo, ro = O.icp_step(Rc, tc, maps[3][0], maps[3][1], O.mat33_inverse(Rp), tp, intr, vprev, nprev, 0.10, th, 0)
    Ar, br, rr = R.icp_step(Rc, tc, maps[3][0], maps[3][1], O.mat33_inverse(Rp), tp, intr, vprev, nprev, 0.10, th)
    assert same(Ao, Ar) and same(bo, br) and same(ro, rr), seed
    # extraction of a random box, then a slab clear on both copies
    lo = [int(v) for v in rng.integers(0, N // 2, 3)]
    hi = [int(min(N, l + rng.integers(4, N))) for l in lo]
    if not _extract_geometry_ok(N, lo, hi):
        lo, hi = [0, 0, lo[2]], [N, N, hi[2]]   # a z range of the whole cross-section: the full grid
    real = [int(v) for v in rng.integers(-3 * N, 3 * N, 3)]
    sub = int(rng.choice([1, 1, 2]))
    po = O.extract_cloud_slice(vo, [size] * 3, 600000, wrap, co, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], sub, real)
    pr = R.extract_cloud_slice(vr, [size] * 3, 600000, wrap, cr, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], sub, real)
    assert len(po) == len(pr) and _point_set(po) == _point_set(pr), (seed, len(po), len(pr))
    if N % 32:
        return   # the reference's clear kernels have no bounds checks (its VOLUME is 512): at a side length that is no multiple of their
                 # block shape they write past the volume (the emulator's guard zones report it); the oracle and the HIP kernel are defined there
    axis, back = int(rng.integers(0, 3)), bool(rng.integers(0, 2))
    cur = int(rng.integers(-2 * N, 2 * N))
    delta = cur + int(rng.integers(1, 20)) * (-1 if back else 1)
    for vol_o, vol_r in ((vo, vr), (co.view(np.uint32).reshape(N, N, N), cr.view(np.uint32).reshape(N, N, N))):
        O.clear_volume(vol_o, axis, back, cur, delta)
        R.clear_volume(vol_r, axis, back, cur, delta)
        assert same(vol_o, vol_r), (seed, axis, back, cur, delta)


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomized_sweep_image_and_rgbd(oracle_mod, R, seed):
    """Seeded random configurations of the image side: vertex / normal maps under random intrinsics and rigid transforms (large
    rotations, translations of metres), the 2x2 map down-sampling, the view products with random light positions and poses, and the RGB-D
    residual + Jacobian reduction at a random pyramid level with random increments -- bit for bit."""
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O = oracle_mod
    rng = np.random.default_rng(7000 + seed)
    cols, rows = [(160, 120), (200, 150), (136, 104), (320, 240)][seed % 4]
    cam = synth.Camera.small(cols, rows)
    scene = synth.Scene(["room", "wall", "farwall"][seed % 3], seed=1234 + seed)
    base = synth.orbit_trajectory(40)
    i0 = int(rng.integers(0, 38))
    (d0, c0), (d1, c1) = [synth.render(scene, cam, *base[i0 + k], noise_mm=float(rng.choice([0.0, 2.0])), rng=rng) for k in (0, 1)]
    d0, d1 = _holes(d0, rng, 0.02), _holes(d1, rng, float(rng.choice([0.0, 0.05])))
    # intrinsics off the nominal ones (principal point off-centre, unequal focal lengths)
    intr = OIntr(cam.fx * float(rng.uniform(0.8, 1.3)), cam.fy * float(rng.uniform(0.8, 1.3)), cam.cx + float(rng.uniform(-20, 20)), cam.cy + float(rng.uniform(-15, 15)))
    lvl = intr.level(int(rng.integers(0, 3)))
    fo = O.bilateral_filter(d0)
    depth_l = fo
    for _ in range(int(round(np.log2(intr.fx / lvl.fx)))):
        nxt = O.pyr_down(depth_l)
        assert same(nxt, R.pyr_down(depth_l))
        depth_l = nxt
    v = O.create_vmap(lvl, depth_l)
    assert same(v, R.create_vmap(lvl, depth_l))
    n = O.create_nmap(v)
    assert same(n, R.create_nmap(v))
    Rm = random_rotation(rng, 2.5)
    t = rng.uniform(-3, 3, 3).astype(np.float32)
    a, b = O.transform_maps(v, n, Rm, t)
    c, d = R.transform_maps(v, n, Rm, t)
    assert same(a, c) and same(b, d)
    if v.shape[1] % 2 == 0 and (v.shape[0] // 3) % 2 == 0:
        assert same(O.resize_map(a, False), R.resize_map(a, False)) and same(O.resize_map(b, True), R.resize_map(b, True))
    # view products from the transformed maps
    rows_l, cols_l = v.shape[0] // 3, v.shape[1]
    colimg = rng.integers(0, 256, (rows_l, cols_l, 4)).astype(np.uint8)
    light = rng.uniform(-20, 20, 3).astype(np.float32)
    ia, ib = O.generate_image(a, b, colimg, light)
    ic, id_ = R.generate_image(a, b, colimg, light)
    assert same(ia, ic) and same(ib, id_)
    Rinv = O.mat33_inverse(random_rotation(rng, 0.3))
    tt = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
    # (the camera-frame maps: a depth behind the camera or beyond 65.5 m converts to unsigned short differently on x86 than CUDA's saturating cvt)
    assert same(O.generate_depth(Rinv, tt, v, n), R.generate_depth(Rinv, tt, v, n, 6.0))
    # RGB-D pyramids, residual and step at a random level
    level = int(rng.integers(0, 3))
    pyr = []
    for dd, rgb in ((d0, c0), (d1, c1)):
        dm, it = O.depth_to_metres(dd, 6000), O.bgr_to_intensity(rgb)
        assert same(dm, R.depth_to_metres(dd, 6000)) and same(it, R.bgr_to_intensity(rgb))
        for _ in range(level):
            if dm.shape[0] % 2 or dm.shape[1] % 2:
                break
            dm2, it2 = O.pyr_down_gauss_f32(dm), O.pyr_down_gauss_u8(it)
            assert same(dm2, R.pyr_down_gauss_f32(dm)) and same(it2, R.pyr_down_gauss_u8(it))
            dm, it = dm2, it2
        pyr.append((dm, it))
    (ld, li), (nd, ni) = pyr
    if ld.shape != nd.shape:
        return
    f = ld.shape[1] / float(cols)
    fx, fy, cx, cy = intr.fx * f, intr.fy * f, intr.cx * f, intr.cy * f
    dx, dy = O.derivative_images(ni)
    dxr, dyr = R.derivative_images(ni)
    assert same(dx, dxr) and same(dy, dyr)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    Rinc = O.rodrigues(rng.uniform(-0.01, 0.01, 3))
    krkinv = (K @ Rinc @ np.linalg.inv(K)).astype(np.float32)
    kt = (K @ rng.uniform(-0.01, 0.01, 3)).astype(np.float32)
    min_scale = (np.float32(rng.choice([12, 5, 3, 1])) / np.float32(0.125)) ** 2
    co_, so, no = O.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, 0.07, kt, krkinv)
    cr_, sr, nr = R.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, 0.07, kt, krkinv)
    assert (so, no) == (sr, nr), seed
    assert np.array_equal(co_["valid"], cr_["valid"] != 0)
    m = co_["valid"] != 0
    for f_ in ("zero", "one", "diff"):
        assert same(co_[f_][m], cr_[f_][m])
    if no == 0:
        return
    cloud = O.project_to_cloud(ld, fx, fy, cx, cy, 0)
    assert same(cloud, R.project_to_cloud(ld, fx, fy, cx, cy, 0))
    sigma = float(np.sqrt(np.float32(no))) if rng.integers(0, 2) else -1.0
    Ao, bo = O.rgb_step(co_, sigma, cloud, fx, fy, dx, dy, 0.125, 0)
    Ar, br = R.rgb_step(cr_, sigma, cloud, fx, fy, dx, dy, 0.125)
    assert same(Ao, Ar) and same(bo, br), (seed, Ao - Ar, bo - br)


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_state_integrate(oracle_mod, R, seed):
    """integrate into volumes in random states (see conftest.random_volume_state; both the reachable and the unrestricted draw): oracle == reference,
    every tsdf word and colour byte."""
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O = oracle_mod
    rng = np.random.default_rng(9000 + seed)
    cols, rows = [(160, 120), (200, 150)][seed % 2]
    N = int(rng.choice([64, 72, 96]))
    size = float(rng.choice([4.0, 6.0]))
    cam = synth.Camera.small(cols, rows)
    scene = synth.Scene(["room", "farwall", "wall"][seed % 3], seed=77 + seed)
    base = synth.orbit_trajectory(40)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    trunc = max(0.06 if size == 6.0 else max(0.01, size / 100), 2.1 * size / N)
    vo, co = random_volume_state(rng, N, reachable=bool(seed % 2))
    vr, cr = vo.copy(), co.copy()
    wrap = [int(v) for v in rng.integers(0, N, 3)]
    for k in range(3):
        Rm, c0 = base[int(rng.integers(0, 40))]
        d, c = synth.render(scene, cam, Rm, c0, noise_mm=1.5, rng=rng)
        c = rng.integers(0, 256, c.shape).astype(np.uint8) if k == 0 else c     # random pixel colours against random stored ones
        Rk = (random_rotation(rng, 0.3) @ np.asarray(Rm, np.float32)).astype(np.float32)
        tk = (np.asarray(c0, np.float32) + np.float32(size / 2) + rng.uniform(-0.2, 0.2, 3)).astype(np.float32)
        n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
        Rinv = O.mat33_inverse(Rk)
        angle = bool(rng.integers(0, 2))
        U, so = O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, angle)
        sr = R.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vr, wrap, cr, c, n, angle)
        assert U > 4000
        assert same(so, sr) and same(vo, vr) and same(co, cr), (seed, k, int((vo != vr).sum()), int((co != cr).any(axis=-1).sum()))


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_state_raycast_and_extract(oracle_mod, R, seed):
    """raycast and extraction on volumes in random states (conftest.random_volume_state with one frame integrated on top, so that rays meet
    zero crossings between arbitrary pairs of stored values, weights and colours -- the trilinear interpolation, the crossing refinement
    and the colour / heat bytes see value combinations a volume grown from empty never holds): oracle == reference."""
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O = oracle_mod
    rng = np.random.default_rng(11000 + seed)
    cols, rows = [(160, 120), (136, 104)][seed % 2]
    N = int(rng.choice([48, 64]))
    size = float(rng.choice([4.0, 6.0]))
    cam = synth.Camera.small(cols, rows)
    scene = synth.Scene(["room", "wall"][seed % 2], seed=3 + seed)
    base = synth.orbit_trajectory(40)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    trunc = max(0.06 if size == 6.0 else max(0.01, size / 100), 2.1 * size / N)
    vo, co = random_volume_state(rng, N, reachable=bool(seed % 2))
    if seed % 4 < 2:
        vo[rng.random((N, N, N)) < 0.5] = 32767        # sparser: rays travel before they meet a crossing
    wrap = [int(v) for v in rng.integers(0, N, 3)]
    Rm, c0 = base[int(rng.integers(0, 40))]
    d, c = synth.render(scene, cam, Rm, c0, noise_mm=1.5, rng=rng)
    Rk = (random_rotation(rng, 0.3) @ np.asarray(Rm, np.float32)).astype(np.float32)
    tk = (np.asarray(c0, np.float32) + np.float32(size / 2) + rng.uniform(-0.2, 0.2, 3)).astype(np.float32)
    n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
    O.integrate_tsdf(d, intr, [size] * 3, O.mat33_inverse(Rk), tk, trunc, vo, wrap, co, c, n, True)
    vr, cr = vo.copy(), co.copy()
    cell = np.float32(size / N)
    hits = 0
    for view in range(8):
        if view >= 3 and hits > 500:   # (a camera inside a negative voxel with positive neighbours sees nothing: look again)
            break
        Rq = (random_rotation(rng, 0.6) @ Rk).astype(np.float32)
        tq = (tk + rng.uniform(-0.5, 0.5, 3)).astype(np.float32)
        outs = []
        for M, vol, col in ((O, vo, co), (R, vr, cr)):
            vm, nm = np.full((3 * rows, cols), 7.0, np.float32), np.full((3 * rows, cols), -3.0, np.float32)
            cm = np.full((rows, cols, 4), 9, np.uint8)
            M.raycast(intr, Rq, tq, trunc, [size] * 3, vol, vm, nm, wrap, cm, col)
            outs.append((vm, nm, cm))
        (a, b, c_), (a2, b2, c2) = outs
        hits += int(np.isfinite(a[:rows]).sum())   # (a camera inside a negative voxel with positive neighbours sees nothing)
        assert same(a, a2) and same(b, b2) and same(c_[..., :3], c2[..., :3]), seed
        hit = np.isfinite(a[:rows])
        g = np.floor(np.stack([a[:rows], a[rows: 2 * rows], a[2 * rows:]], -1) / cell)
        border = hit & ((g <= 0) | (g >= N - 1)).any(axis=-1)
        assert same(c_[..., 3][~border], c2[..., 3][~border])
    assert hits > 500
    lo = [int(v) for v in rng.integers(0, N // 2, 3)]
    hi = [int(min(N, l + rng.integers(4, N))) for l in lo]
    if not _extract_geometry_ok(N, lo, hi):
        lo, hi = [0, 0, lo[2]], [N, N, hi[2]]
    real = [int(v) for v in rng.integers(-3 * N, 3 * N, 3)]
    po = O.extract_cloud_slice(vo, [size] * 3, 2000000, wrap, co, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, real)
    pr = R.extract_cloud_slice(vr, [size] * 3, 2000000, wrap, cr, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, real)
    assert len(po) == len(pr) and _point_set(po) == _point_set(pr), (seed, len(po), len(pr))


@pytest.mark.parametrize("seed", list(range(8)))
def test_perturbed_maps_icp(oracle_mod, R, seed):
    """The ICP reduction on perturbed maps (conftest.perturbed_maps: random vertices, un-normalised normals and NaN holes in both the
    current and the predicted maps) at random relative poses, including large ones: A, b and the residual pair bit for bit."""
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    from conftest import perturbed_maps
    O = oracle_mod
    rng = np.random.default_rng(13000 + seed)
    cols, rows = [(160, 120), (200, 150), (320, 240)][seed % 3]
    cam = synth.Camera.small(cols, rows)
    scene = synth.Scene(["room", "wall", "farwall"][seed % 3], seed=9 + seed)
    base = synth.orbit_trajectory(40)
    i0 = int(rng.integers(0, 38))
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    ms = []
    for k in (0, 1):
        d, _ = synth.render(scene, cam, *base[i0 + k], noise_mm=1.0, rng=rng)
        v = O.create_vmap(intr, O.bilateral_filter(d))
        ms.append(perturbed_maps(rng, v, O.create_nmap(v), float(rng.choice([0.05, 0.3]))))
    (vc, nc), (vp, npv) = ms
    Rp = random_rotation(rng, 0.4)
    tp = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
    vg, ng = O.transform_maps(vp, npv, Rp, tp)      # the prediction lives in the global frame
    for mag in (0.01, 0.2):
        Rc = (random_rotation(rng, mag) @ Rp).astype(np.float32)
        tc = (tp + rng.uniform(-mag, mag, 3)).astype(np.float32)
        dist, th = float(rng.choice([0.10, 0.5, 5.0])), float(np.sin(np.float32(rng.choice([20.0, 60.0]) * 3.14159265 / 180.0)))
        Ao, bo, ro = O.icp_step(Rc, tc, vc, nc, O.mat33_inverse(Rp), tp, intr, vg, ng, dist, th, 0)
        Ar, br, rr = R.icp_step(Rc, tc, vc, nc, O.mat33_inverse(Rp), tp, intr, vg, ng, dist, th)
        assert np.asarray(ro).ravel()[1] > 50 or mag > 0.1, (seed, ro)
        assert same(Ao, Ar) and same(bo, br) and same(ro, rr), (seed, mag)


@pytest.mark.parametrize("seed", list(range(6)))
def test_noise_images_rgbd(oracle_mod, R, seed):
    """The photometric side on NOISE: intensity images of random bytes (every pixel passes the gradient threshold, every 4x4 window is
    non-zero or not at random), depths with random holes and steps, large increments -- residual search, sigma, count and the Jacobian
    reduction, bit for bit."""
    O = oracle_mod
    rng = np.random.default_rng(15000 + seed)
    cols, rows = [(160, 120), (80, 60), (320, 240)][seed % 3]
    fx = fy = 528.0 * cols / 640
    cx, cy = cols / 2 - 0.5 + float(rng.uniform(-5, 5)), rows / 2 - 0.5 + float(rng.uniform(-5, 5))
    imgs, deps = [], []
    for k in range(2):
        it = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
        if seed % 2:
            it = np.repeat(np.repeat(rng.integers(0, 256, (rows // 4, cols // 4)).astype(np.uint8), 4, 0), 4, 1)   # 4x4 plateaus: zero gradients too
        it[rng.random((rows, cols)) < 0.1] = 0
        dm = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        dm = np.where(rng.random((rows, cols)) < 0.7, np.float32(1.5) + np.float32(0.3) * np.sin(np.arange(cols, dtype=np.float32) / 9)[None, :], dm).astype(np.float32)
        dm[rng.random((rows, cols)) < 0.1] = 0
        if seed == 5:
            dm[rng.random((rows, cols)) < 0.02] = np.nan
        imgs.append(it); deps.append(dm)
    (li, ni), (ld, nd) = imgs, deps
    dx, dy = O.derivative_images(ni)
    dxr, dyr = R.derivative_images(ni)
    assert same(dx, dxr) and same(dy, dyr)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    for mag in (0.002, 0.05):
        Rinc = O.rodrigues(rng.uniform(-mag, mag, 3))
        krkinv = (K @ Rinc @ np.linalg.inv(K)).astype(np.float32)
        kt = (K @ rng.uniform(-mag, mag, 3)).astype(np.float32)
        min_scale = (np.float32(rng.choice([12, 5, 3, 1])) / np.float32(0.125)) ** 2
        delta = float(rng.choice([0.07, 1.0]))
        co_, so, no = O.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, delta, kt, krkinv)
        cr_, sr, nr = R.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, delta, kt, krkinv)
        assert (so, no) == (sr, nr), (seed, mag, so, sr, no, nr)
        assert np.array_equal(co_["valid"], cr_["valid"] != 0)
        m = co_["valid"] != 0
        for f_ in ("zero", "one", "diff"):
            assert same(co_[f_][m], cr_[f_][m])
        if no == 0:
            continue
        cloud = O.project_to_cloud(ld, fx, fy, cx, cy, 0)
        assert same(cloud, R.project_to_cloud(ld, fx, fy, cx, cy, 0))
        sigma = float(np.sqrt(np.float32(no))) if rng.integers(0, 2) else -1.0
        Ao, bo = O.rgb_step(co_, sigma, cloud, fx, fy, dx, dy, 0.125, 0)
        Ar, br = R.rgb_step(cr_, sigma, cloud, fx, fy, dx, dy, 0.125)
        assert same(Ao, Ar) and same(bo, br), (seed, mag)


def test_full_size_frames(oracle_mod, R):
    """The bench's own configuration -- 640x480 frames of the orbit sequence into a 512^3 volume with a storage wrap, the raycast from the
    next pose, an ICP reduction and the RGB-D residual + step at full resolution, the whole volume extracted -- through tests/tools/full_size_pin.py (its own process:
    two volume pairs are 1.6 GB).  The tool has been run over 40 frames (160 M voxel updates); two here."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "full_size_pin.py")
    r = subprocess.run([sys.executable, tool, "2", "orbit512", "quick"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


# the emulated kernels make a frame cost seconds: the suite keeps two short runs (one of them through a volume shift), KT_DEEP_PIN=1 the
# full set (11 minutes; run for round 2: all identical)
_TRACKER_RUNS = [("icp", 70), ("rgbd_icp", 36), ("rgbd", 24), ("fast_odometry", 40), ("dynamic_cube", 30), ("static", 8)] \
    if os.environ.get("KT_DEEP_PIN") else [("icp", 28), ("static", 5)]


@pytest.mark.parametrize("mode,frames", _TRACKER_RUNS)
def test_tracker_on_reference_kernels(oracle_mod, R, tmp_path, mode, frames):
    """Whole runs instead of constructed inputs: the oracle's tracker (processFrame state machine, Gauss-Newton loops, volume shifts,
    place-recognition tap) once on the oracle's kernels and once with EVERY kernel call rerouted to the reference's own kernels
    (oracle/_ref/libkt_oracle_on_ref.so, ref_shim/kt_oracle_on_ref.h; the bilateral filter aside, whose exp() model is the documented
    difference).  Every pose, the wraps along the way, the final volumes, every slice and the last prediction: bit for bit."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    tool = os.path.join(here, "tools", "tracker_dump.py")
    hybrid = os.path.join(os.path.dirname(here), "oracle", "_ref", "libkt_oracle_on_ref.so")
    subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(here), "oracle"), "_ref/libkt_oracle_on_ref.so"], stdout=subprocess.DEVNULL)
    outs = []
    for name, lib in (("oracle", None), ("on_ref", hybrid)):
        env = dict(os.environ)
        env.pop("KT_ORACLE_LIB", None)
        if lib:
            env["KT_ORACLE_LIB"] = lib
        out = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, tool, mode, str(frames), out], env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(dict(np.load(out)))
    a, b = outs
    assert set(a) == set(b)
    if mode in ("icp", "rgbd_icp", "fast_odometry"):
        assert len(a["slice_sizes"]) >= 2 and int(np.abs(a["wraps"][-1]).sum()) > 0, a["wraps"][-1]     # the run did shift
    for k in a:
        assert a[k].shape == b[k].shape and np.array_equal(np.ascontiguousarray(a[k]).view(np.uint8), np.ascontiguousarray(b[k]).view(np.uint8)), (mode, k)
