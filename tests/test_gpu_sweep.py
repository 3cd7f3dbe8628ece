"""GPU parity over seeded random configurations: the HIP kernels (through the C-ABI) against the oracle on the configurations the CPU
sweep of tests/test_oracle_vs_ref.py runs against the reference's own kernels -- scene, image size (no multiples of tile or block
shapes), volume side and extent, storage wrap, poses with large rotations, holes and noise, the colour-angle flag; integrate into one
volume over 4 frames, raycast from 2 poses, extraction of a random box with real voxel wraps of a few hundred voxels (where the
offset arithmetic of the output points is sensitive to contraction), a slab clear, one ICP reduction.  Bit for bit (NaN payloads aside)."""
import numpy as np
import pytest

from conftest import random_rotation

pytestmark = pytest.mark.gpu


def _holes(depth, rng, frac):
    d = depth.copy()
    d[rng.random(d.shape) < frac] = 0
    return d


def _same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _same_maps(a, b):
    """Float maps: the same NaN positions (a NaN's sign and payload are not specified by either side: 0/0 in the normal of a hit next
    to a volume face gives +NaN on the host and -NaN on the GPU) and the same bits everywhere else."""
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def _point_set(p):
    return sorted(bytes(r) for r in np.ascontiguousarray(p).view(np.uint8).reshape(len(p), -1)) if len(p) else []


@pytest.mark.parametrize("seed", list(range(16)))
def test_randomized_sweep_hip(ctx, oracle_mod, seed):
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
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
    vh, ch = vo.copy(), co.copy()
    poses, maps = [], []
    for k in range(4):
        Rm, c0 = base[int(rng.integers(0, 40))]
        d, c = synth.render(scene, cam, Rm, c0, noise_mm=float(rng.choice([0.0, 1.5])), rng=rng)
        d = _holes(d, rng, float(rng.choice([0.0, 0.02])))
        Rk = (random_rotation(rng, 0.5) @ np.asarray(Rm, np.float32)).astype(np.float32)
        tk = (np.asarray(c0, np.float32) + np.float32(size / 2) + rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
        fo, fh = O.bilateral_filter(d), H.bilateral_filter(d)
        assert _same(fo, fh), (seed, k, "bilateral")
        v, vv = O.create_vmap(intr, fo), H.create_vmap(intr, fo)
        assert _same(v, vv)
        n, nn = O.create_nmap(v), H.create_nmap(v)
        assert _same(n, nn)
        Rinv = O.mat33_inverse(Rk)
        U, so = O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, angle)
        sh = H.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vh, wrap, ch, c, n, angle)
        assert _same(so, sh) and _same(vo, vh) and _same(co, ch), (seed, k, int((vo != vh).sum()), int((co != ch).any(axis=-1).sum()))
        poses.append((Rk, tk))
        maps.append((v, n))
    pred = None
    for Rk, tk in poses[2:]:
        outs = []
        for M in (O, H):
            vm, nm = np.full((3 * rows, cols), 7.0, np.float32), np.full((3 * rows, cols), -3.0, np.float32)
            cm = np.full((rows, cols, 4), 9, np.uint8)
            M.raycast(intr, Rk, tk, trunc, [size] * 3, vo, vm, nm, wrap, cm, co)
            outs.append((vm, nm, cm))
        (a, b, c_), (a2, b2, c2) = outs
        assert _same_maps(a, a2) and _same_maps(b, b2) and _same(c_, c2), (seed, "raycast")
        pred = (Rk, tk, a, b)
    Rp, tp, vprev, nprev = pred
    Rc = (random_rotation(rng, 0.03) @ Rp).astype(np.float32)
    tc = (tp + rng.uniform(-0.02, 0.02, 3)).astype(np.float32)
    th = float(np.sin(np.float32(20.0 * 3.14159265 / 180.0)))
    Ao, bo, ro = O.icp_step(Rc, tc, maps[3][0], maps[3][1], O.mat33_inverse(Rp), tp, intr, vprev, nprev, 0.10, th, 0)
    Ah, bh, rh = H.icp_step(Rc, tc, maps[3][0], maps[3][1], O.mat33_inverse(Rp), tp, intr, vprev, nprev, 0.10, th)
    assert _same(np.asarray(Ao, np.float32), np.asarray(Ah, np.float32)) and _same(np.asarray(bo, np.float32), np.asarray(bh, np.float32)), seed
    assert _same(np.asarray(ro, np.float32), np.asarray(rh, np.float32)), seed
    lo = [int(v) for v in rng.integers(0, N // 2, 3)]
    hi = [int(min(N, l + rng.integers(4, N))) for l in lo]
    real = [int(v) for v in rng.integers(-3 * N, 3 * N, 3)]
    sub = int(rng.choice([1, 1, 2]))
    po = O.extract_cloud_slice(vo, [size] * 3, 600000, wrap, co, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], sub, real)
    ph = H.extract_cloud_slice(vh, [size] * 3, 600000, wrap, ch, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], sub, real)
    assert len(po) == len(ph) and _point_set(po) == _point_set(ph), (seed, len(po), len(ph))
    axis, back = int(rng.integers(0, 3)), bool(rng.integers(0, 2))
    cur = int(rng.integers(-2 * N, 2 * N))
    delta = cur + int(rng.integers(1, 20)) * (-1 if back else 1)
    for vol_o, vol_h in ((vo, vh), (co.view(np.uint32).reshape(N, N, N), ch.view(np.uint32).reshape(N, N, N))):
        O.clear_volume(vol_o, axis, back, cur, delta)
        H.clear_volume(vol_h, axis, back, cur, delta)
        assert _same(vol_o, vol_h), (seed, axis, back, cur, delta)


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomized_sweep_hip_image_and_rgbd(ctx, oracle_mod, seed):
    """The image side over seeded random configurations (the CPU counterpart runs against the reference's kernels): maps under random
    intrinsics and rigid transforms, 2x2 down-sampling, RGB-D pyramids, derivative images, point clouds, residual and Jacobian reduction."""
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
    rng = np.random.default_rng(7000 + seed)
    cols, rows = [(160, 120), (200, 150), (136, 104), (320, 240)][seed % 4]
    cam = synth.Camera.small(cols, rows)
    scene = synth.Scene(["room", "wall", "farwall"][seed % 3], seed=1234 + seed)
    base = synth.orbit_trajectory(40)
    i0 = int(rng.integers(0, 38))
    (d0, c0), (d1, c1) = [synth.render(scene, cam, *base[i0 + k], noise_mm=float(rng.choice([0.0, 2.0])), rng=rng) for k in (0, 1)]
    d0, d1 = _holes(d0, rng, 0.02), _holes(d1, rng, float(rng.choice([0.0, 0.05])))
    intr = OIntr(cam.fx * float(rng.uniform(0.8, 1.3)), cam.fy * float(rng.uniform(0.8, 1.3)), cam.cx + float(rng.uniform(-20, 20)), cam.cy + float(rng.uniform(-15, 15)))
    lvl = intr.level(int(rng.integers(0, 3)))
    fo = O.bilateral_filter(d0)
    assert _same(fo, H.bilateral_filter(d0))
    noisy = rng.integers(0, 65536, d0.shape).astype(np.uint16)
    assert _same(O.bilateral_filter(noisy), H.bilateral_filter(noisy))
    depth_l = fo
    for _ in range(int(round(np.log2(intr.fx / lvl.fx)))):
        nxt = O.pyr_down(depth_l)
        assert _same(nxt, H.pyr_down(depth_l))
        depth_l = nxt
    v = O.create_vmap(lvl, depth_l)
    assert _same_maps(v, H.create_vmap(lvl, depth_l))
    n = O.create_nmap(v)
    assert _same_maps(n, H.create_nmap(v))
    Rm = random_rotation(rng, 2.5)
    t = rng.uniform(-3, 3, 3).astype(np.float32)
    a, b = O.transform_maps(v, n, Rm, t)
    c, d = H.transform_maps(v, n, Rm, t)
    assert _same_maps(a, c) and _same_maps(b, d)
    if v.shape[1] % 2 == 0 and (v.shape[0] // 3) % 2 == 0:
        assert _same_maps(O.resize_map(a, False), H.resize_map(a, False)) and _same_maps(O.resize_map(b, True), H.resize_map(b, True))
    level = int(rng.integers(0, 3))
    pyr = []
    for dd, rgb in ((d0, c0), (d1, c1)):
        dm, it = O.depth_to_metres(dd, 6000), O.bgr_to_intensity(rgb)
        assert _same(dm, H.depth_to_metres(dd, 6000)) and _same(it, H.bgr_to_intensity(rgb))
        for _ in range(level):
            if dm.shape[0] % 2 or dm.shape[1] % 2:
                break
            dm2, it2 = O.pyr_down_gauss_f32(dm), O.pyr_down_gauss_u8(it)
            assert _same(dm2, H.pyr_down_gauss_f32(dm)) and _same(it2, H.pyr_down_gauss_u8(it))
            dm, it = dm2, it2
        pyr.append((dm, it))
    (ld, li), (nd, ni) = pyr
    if ld.shape != nd.shape:
        return
    f = ld.shape[1] / float(cols)
    fx, fy, cx, cy = intr.fx * f, intr.fy * f, intr.cx * f, intr.cy * f
    dx, dy = O.derivative_images(ni)
    dxh, dyh = H.derivative_images(ni)
    assert _same(dx, dxh) and _same(dy, dyh)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    Rinc = O.rodrigues(rng.uniform(-0.01, 0.01, 3))
    krkinv = (K @ Rinc @ np.linalg.inv(K)).astype(np.float32)
    kt = (K @ rng.uniform(-0.01, 0.01, 3)).astype(np.float32)
    min_scale = (np.float32(rng.choice([12, 5, 3, 1])) / np.float32(0.125)) ** 2
    co_, so, no = O.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, 0.07, kt, krkinv)
    ch_, sh, nh = H.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, 0.07, kt, krkinv)
    assert (so, no) == (sh, nh), seed
    assert np.array_equal(co_["valid"] != 0, ch_["valid"] != 0)
    m = co_["valid"] != 0
    for f_ in ("zero", "one", "diff"):
        assert _same(co_[f_][m], ch_[f_][m])
    if no == 0:
        return
    cloud = O.project_to_cloud(ld, fx, fy, cx, cy, 0)
    assert _same_maps(cloud, H.project_to_cloud(ld, fx, fy, cx, cy, 0))
    sigma = float(np.sqrt(np.float32(no))) if rng.integers(0, 2) else -1.0
    Ao, bo = O.rgb_step(co_, sigma, cloud, fx, fy, dx, dy, 0.125, 0)
    Ah, bh = H.rgb_step(ch_, sigma, cloud, fx, fy, dx, dy, 0.125)
    assert _same(np.asarray(Ao, np.float32), np.asarray(Ah, np.float32)) and _same(np.asarray(bo, np.float32), np.asarray(bh, np.float32)), seed


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_state_integrate_hip(ctx, oracle_mod, seed):
    """integrate into volumes in random STATES (conftest.random_volume_state, the draw restricted to states the kernels can leave behind:
    weights <= 128, colour 0 where the weight is 0 -- the two invariants the HIP kernel's weight clamp and same-colour shortcut lean on,
    kt_abi.h): a few hundred thousand (stored value, weight, pixel colour, colour weight) combinations per call through the running
    averages and their quantisation.  HIP == oracle, every tsdf word and colour byte."""
    from conftest import random_volume_state
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
    rng = np.random.default_rng(9000 + seed)
    cols, rows = [(160, 120), (200, 150)][seed % 2]
    N = int(rng.choice([64, 72, 96]))
    size = float(rng.choice([4.0, 6.0]))
    cam = synth.Camera.small(cols, rows)
    scene = synth.Scene(["room", "farwall", "wall"][seed % 3], seed=77 + seed)
    base = synth.orbit_trajectory(40)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    trunc = max(0.06 if size == 6.0 else max(0.01, size / 100), 2.1 * size / N)
    vo, co = random_volume_state(rng, N, reachable=bool(seed % 2))   # (the weight update saturates and clamps like tsdf_volume.cu:621 for any byte)
    vh, ch = vo.copy(), co.copy()
    wrap = [int(v) for v in rng.integers(0, N, 3)]
    for k in range(3):
        Rm, c0 = base[int(rng.integers(0, 40))]
        d, c = synth.render(scene, cam, Rm, c0, noise_mm=1.5, rng=rng)
        c = rng.integers(0, 256, c.shape).astype(np.uint8) if k == 0 else c
        Rk = (random_rotation(rng, 0.3) @ np.asarray(Rm, np.float32)).astype(np.float32)
        tk = (np.asarray(c0, np.float32) + np.float32(size / 2) + rng.uniform(-0.2, 0.2, 3)).astype(np.float32)
        n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
        Rinv = O.mat33_inverse(Rk)
        angle = bool(rng.integers(0, 2))
        U, so = O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, angle)
        sh = H.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vh, wrap, ch, c, n, angle)
        assert U > 4000
        bad_v, bad_c = np.argwhere(vo != vh), np.argwhere((co != ch).any(axis=-1))
        assert _same(so, sh) and len(bad_v) == 0 and len(bad_c) == 0, (seed, k, len(bad_v), len(bad_c), bad_v[:3].tolist(), bad_c[:3].tolist())


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_state_raycast_and_extract_hip(ctx, oracle_mod, seed):
    """raycast and extraction on volumes in random states (one frame integrated on top of conftest.random_volume_state): rays meet zero
    crossings between arbitrary pairs of stored values, weights and colours.  HIP == oracle (the CPU twin of this test holds the oracle
    against the reference's kernels on the same cases)."""
    from conftest import random_volume_state
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
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
        vo[rng.random((N, N, N)) < 0.5] = 32767
    wrap = [int(v) for v in rng.integers(0, N, 3)]
    Rm, c0 = base[int(rng.integers(0, 40))]
    d, c = synth.render(scene, cam, Rm, c0, noise_mm=1.5, rng=rng)
    Rk = (random_rotation(rng, 0.3) @ np.asarray(Rm, np.float32)).astype(np.float32)
    tk = (np.asarray(c0, np.float32) + np.float32(size / 2) + rng.uniform(-0.2, 0.2, 3)).astype(np.float32)
    n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
    O.integrate_tsdf(d, intr, [size] * 3, O.mat33_inverse(Rk), tk, trunc, vo, wrap, co, c, n, True)   # (states above weight 128: raycast and extraction only)
    vh, ch = vo.copy(), co.copy()
    hits = 0
    for view in range(8):
        if view >= 3 and hits > 500:   # (a camera inside a negative voxel with positive neighbours sees nothing: look again)
            break
        Rq = (random_rotation(rng, 0.6) @ Rk).astype(np.float32)
        tq = (tk + rng.uniform(-0.5, 0.5, 3)).astype(np.float32)
        outs = []
        for M, vol, col in ((O, vo, co), (H, vh, ch)):
            vm, nm = np.full((3 * rows, cols), 7.0, np.float32), np.full((3 * rows, cols), -3.0, np.float32)
            cm = np.full((rows, cols, 4), 9, np.uint8)
            M.raycast(intr, Rq, tq, trunc, [size] * 3, vol, vm, nm, wrap, cm, col)
            outs.append((vm, nm, cm))
        (a, b, c_), (a2, b2, c2) = outs
        hits += int(np.isfinite(a[:rows]).sum())
        assert _same_maps(a, a2) and _same_maps(b, b2) and _same(c_, c2), (seed, "raycast", int((a.view(np.uint32) != a2.view(np.uint32)).sum()),
                                                                           int((b.view(np.uint32) != b2.view(np.uint32)).sum()), int((c_ != c2).sum()))
    assert hits > 500
    lo = [int(v) for v in rng.integers(0, N // 2, 3)]
    hi = [int(min(N, l + rng.integers(4, N))) for l in lo]
    real = [int(v) for v in rng.integers(-3 * N, 3 * N, 3)]
    po = O.extract_cloud_slice(vo, [size] * 3, 2000000, wrap, co, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, real)
    ph = H.extract_cloud_slice(vh, [size] * 3, 2000000, wrap, ch, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, real)
    assert len(po) == len(ph) and _point_set(po) == _point_set(ph), (seed, len(po), len(ph))


@pytest.mark.parametrize("seed", list(range(8)))
def test_perturbed_maps_icp_hip(ctx, oracle_mod, seed):
    """The ICP reduction on perturbed maps (conftest.perturbed_maps: random vertices, un-normalised normals, NaN holes) at small and large
    relative poses and several thresholds: HIP == oracle in A, b and the residual pair (CPU twin: oracle == reference)."""
    from conftest import perturbed_maps
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
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
    vg, ng = O.transform_maps(vp, npv, Rp, tp)
    for mag in (0.01, 0.2):
        Rc = (random_rotation(rng, mag) @ Rp).astype(np.float32)
        tc = (tp + rng.uniform(-mag, mag, 3)).astype(np.float32)
        dist, th = float(rng.choice([0.10, 0.5, 5.0])), float(np.sin(np.float32(rng.choice([20.0, 60.0]) * 3.14159265 / 180.0)))
        Ao, bo, ro = O.icp_step(Rc, tc, vc, nc, O.mat33_inverse(Rp), tp, intr, vg, ng, dist, th, 0)
        Ah, bh, rh = H.icp_step(Rc, tc, vc, nc, O.mat33_inverse(Rp), tp, intr, vg, ng, dist, th)
        f = lambda x: np.asarray(x, np.float32)
        assert _same_maps(f(Ao), f(Ah)) and _same_maps(f(bo), f(bh)) and _same_maps(f(ro), f(rh)), (seed, mag, f(ro), f(rh))


@pytest.mark.parametrize("seed", list(range(6)))
def test_noise_images_rgbd_hip(ctx, oracle_mod, seed):
    """The photometric side on noise images (random bytes or 4x4 plateaus, zeroed pixels), depths with holes, steps and -- in one case --
    NaNs, small and large increments: residual search, sigma, count and the Jacobian reduction, HIP == oracle (CPU twin: oracle ==
    reference)."""
    from hip_kernels import HipKernels
    O, H = oracle_mod, HipKernels(ctx)
    rng = np.random.default_rng(15000 + seed)
    cols, rows = [(160, 120), (80, 60), (320, 240)][seed % 3]
    fx = fy = 528.0 * cols / 640
    cx, cy = cols / 2 - 0.5 + float(rng.uniform(-5, 5)), rows / 2 - 0.5 + float(rng.uniform(-5, 5))
    imgs, deps = [], []
    for k in range(2):
        it = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
        if seed % 2:
            it = np.repeat(np.repeat(rng.integers(0, 256, (rows // 4, cols // 4)).astype(np.uint8), 4, 0), 4, 1)
        it[rng.random((rows, cols)) < 0.1] = 0
        dm = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        dm = np.where(rng.random((rows, cols)) < 0.7, np.float32(1.5) + np.float32(0.3) * np.sin(np.arange(cols, dtype=np.float32) / 9)[None, :], dm).astype(np.float32)
        dm[rng.random((rows, cols)) < 0.1] = 0
        if seed == 5:
            dm[rng.random((rows, cols)) < 0.02] = np.nan
        imgs.append(it); deps.append(dm)
    (li, ni), (ld, nd) = imgs, deps
    dx, dy = O.derivative_images(ni)
    dxh, dyh = H.derivative_images(ni)
    assert _same(dx, dxh) and _same(dy, dyh)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    for mag in (0.002, 0.05):
        Rinc = O.rodrigues(rng.uniform(-mag, mag, 3))
        krkinv = (K @ Rinc @ np.linalg.inv(K)).astype(np.float32)
        kt = (K @ rng.uniform(-mag, mag, 3)).astype(np.float32)
        min_scale = (np.float32(rng.choice([12, 5, 3, 1])) / np.float32(0.125)) ** 2
        delta = float(rng.choice([0.07, 1.0]))
        co_, so, no = O.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, delta, kt, krkinv)
        ch_, sh, nh = H.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, delta, kt, krkinv)
        assert (so, no) == (sh, nh), (seed, mag, so, sh, no, nh)
        assert np.array_equal(co_["valid"] != 0, ch_["valid"] != 0)
        m = co_["valid"] != 0
        for f_ in ("zero", "one", "diff"):
            assert _same(co_[f_][m], ch_[f_][m])
        if no == 0:
            continue
        cloud = O.project_to_cloud(ld, fx, fy, cx, cy, 0)
        assert _same_maps(cloud, H.project_to_cloud(ld, fx, fy, cx, cy, 0))
        sigma = float(np.sqrt(np.float32(no))) if rng.integers(0, 2) else -1.0
        Ao, bo = O.rgb_step(co_, sigma, cloud, fx, fy, dx, dy, 0.125, 0)
        Ah, bh = H.rgb_step(ch_, sigma, cloud, fx, fy, dx, dy, 0.125)
        f = lambda x: np.asarray(x, np.float32)
        assert _same_maps(f(Ao), f(Ah)) and _same_maps(f(bo), f(bh)), (seed, mag)
