"""Known-answer tests that pin the CPU oracle (the reference ships no tests or golden vectors; the kernels are
also pinned against the reference's own sources in test_oracle_vs_ref.py -- these tests are the only pin of the HOST logic).
Analytic checks only: what the reference's algorithm must produce on inputs whose answer is known in closed form."""
import math

import numpy as np
import pytest

from oracle.oracle import OIntr


def test_float_to_int_conventions(oracle_mod):
    L = oracle_mod.lib()
    # CUDA __float2int_rn: ties to even, saturation, NaN -> 0
    for x, rn, rz, rd in [(0.5, 0, 0, 0), (1.5, 2, 1, 1), (2.5, 2, 2, 2), (-0.5, 0, 0, -1), (-1.5, -2, -1, -2), (3.49, 3, 3, 3),
                          (float("nan"), 0, 0, 0), (3e9, 2**31 - 1, 2**31 - 1, 2**31 - 1), (-3e9, -2**31, -2**31, -2**31),
                          (float("inf"), 2**31 - 1, 2**31 - 1, 2**31 - 1)]:
        assert L.kto_f2i_rn(x) == rn and L.kto_f2i_rz(x) == rz and L.kto_f2i_rd(x) == rd, x


def test_expf_restatement(oracle_mod):
    L = oracle_mod.lib()
    xs = -np.concatenate([np.linspace(0, 86, 4001), np.logspace(-6, 1.9, 300)])
    got = np.array([L.kto_expf(float(x)) for x in xs.astype(np.float32)])
    ref = np.exp(xs.astype(np.float32).astype(np.float64))
    ok = ref > 2.0 ** -124
    assert np.abs(got[ok] / ref[ok] - 1).max() < 3e-7  # ~2 ulp, the accuracy class of __expf
    assert L.kto_expf(0.0) == 1.0
    assert L.kto_expf(-100.0) == 0.0 and L.kto_expf(float("nan")) == 0.0  # flushed like --ftz=true


def test_host_math(oracle_mod):
    rng = np.random.default_rng(0)
    for _ in range(20):
        M = rng.normal(size=(6, 8))
        A = M @ M.T + 1e-3 * np.eye(6)
        b = rng.normal(size=6)
        x = oracle_mod.ldlt_solve6(A, b)
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # rank-deficient system: Eigen's LDLT solves with the pseudo-inverse of D
    A = np.diag([4.0, 0, 1, 0, 2, 3])
    x = oracle_mod.ldlt_solve6(A, np.ones(6))
    assert np.allclose(x, [0.25, 0, 1, 0, 0.5, 1 / 3])
    from scipy.spatial.transform import Rotation
    for _ in range(20):
        r = rng.normal(size=3) * rng.uniform(0, 2)
        assert np.allclose(oracle_mod.rodrigues(r), Rotation.from_rotvec(r).as_matrix(), atol=1e-14)
        R = Rotation.from_rotvec(r).as_matrix().astype(np.float32)
        q = oracle_mod.quat_from_mat33(R)
        qs = Rotation.from_matrix(R.astype(np.float64)).as_quat()
        assert min(np.abs(q - qs).max(), np.abs(q + qs).max()) < 1e-6
        assert np.allclose(oracle_mod.mat33_inverse(R), np.linalg.inv(R.astype(np.float64)), atol=1e-6)
    assert np.array_equal(oracle_mod.rodrigues([0, 0, 0]), np.eye(3))


def _plane_depth(cols, rows, z_mm):
    return np.full((rows, cols), z_mm, np.uint16)


def test_vertex_and_normal_maps_of_a_plane(oracle_mod):
    cols, rows = 64, 48
    intr = OIntr(60.0, 60.0, 32.0, 24.0)
    d = _plane_depth(cols, rows, 2000)
    d[5, 7] = 0
    v = oracle_mod.create_vmap(intr, d)
    n = oracle_mod.create_nmap(v)
    u, w = 20, 30
    assert v[w, u] == pytest.approx(2.0 * (u - 32) / 60, rel=1e-6) and v[rows + w, u] == pytest.approx(2.0 * (w - 24) / 60, rel=1e-6)
    assert v[2 * rows + w, u] == pytest.approx(2.0)
    assert math.isnan(v[5, 7]) and math.isnan(n[5, 7]) and math.isnan(n[5, 6]) and math.isnan(n[4, 7])  # hole poisons its left / upper neighbours
    assert np.isnan(n[:rows, cols - 1]).all() and np.isnan(n[rows - 1, :]).all()  # last column / row (maps.cu:90-94)
    # a fronto-parallel plane has normal (0, 0, 1) (cross of +x and +y edges)
    assert n[w, u] == pytest.approx(0, abs=1e-6) and n[rows + w, u] == pytest.approx(0, abs=1e-6) and n[2 * rows + w, u] == pytest.approx(1.0, abs=1e-6)


def test_bilateral_constant_and_pyramid(oracle_mod):
    d = _plane_depth(64, 48, 1500)
    f = oracle_mod.bilateral_filter(d)
    assert (f == 1500).all()  # weighted mean of a constant; the never-sampled last row / column does not matter
    p = oracle_mod.pyr_down(f)
    assert p.shape == (24, 32) and (p == 1500).all()
    # a depth edge larger than 3 * sigma_color (90 mm) is not averaged across in pyrDown
    d2 = d.copy()
    d2[:, 32:] = 2500
    p2 = oracle_mod.pyr_down(d2)
    assert set(np.unique(p2)) == {1500, 2500}


def test_integrate_plane_gives_linear_tsdf(oracle_mod):
    """Camera at the volume centre looking along +z at a wall 1 m away: F(voxel) = clamp((d_wall - range) / trunc) along the
    optical axis, weight 1 where updated, 0 behind the wall beyond the truncation band (tsdf_volume.cu:596-621)."""
    cols, rows, N, size, trunc = 64, 48, 64, 4.0, 0.25
    intr = OIntr(60.0, 60.0, 32.0, 24.0)
    d = _plane_depth(cols, rows, 1000)
    v = oracle_mod.create_vmap(intr, d)
    n = oracle_mod.create_nmap(v)
    vol = np.zeros((N, N, N), np.int16)
    col = np.zeros((N, N, N, 4), np.uint8)
    rgb = np.full((rows, cols, 3), 200, np.uint8)
    t = np.array([2.0, 2.0, 2.0], np.float32)
    U, scaled = oracle_mod.integrate_tsdf(d, intr, [size] * 3, np.eye(3), t, trunc, vol, [0, 0, 0], col, rgb, n, True)
    cell = size / N
    # the voxel column through the principal point: x, y index whose centre is closest to the optical axis
    ix, iy = int((2.0 + cell * 0.0) / cell), int(2.0 / cell)
    zs = (np.arange(N) + 0.5) * cell - 2.0
    px, py = (ix + 0.5) * cell - 2.0, (iy + 0.5) * cell - 2.0
    for iz in range(N):
        z = zs[iz]
        if z <= 0.05:
            continue
        rng_ = math.sqrt(px * px + py * py + z * z)
        u, w = round(px / z * 60 + 32), round(py / z * 60 + 24)
        lam = math.sqrt(((u - 32) / 60) ** 2 + ((w - 24) / 60) ** 2 + 1)
        sdf = 1.0 * lam - rng_
        if sdf >= -trunc:
            exp = min(1.0, sdf / trunc)
            assert col[iz, iy, ix, 3] == 1
            assert abs(vol[iz, iy, ix] / 32767.0 - exp) < 2e-4, (iz, vol[iz, iy, ix] / 32767.0, exp)
            assert tuple(col[iz, iy, ix, :3]) == (200, 200, 200)
        else:
            assert col[iz, iy, ix, 3] == 0 and vol[iz, iy, ix] == 0
    assert U == int((col[..., 3] == 1).sum())
    # second integration of the same frame: running average keeps F, weight becomes 2
    vol2, col2 = vol.copy(), col.copy()
    oracle_mod.integrate_tsdf(d, intr, [size] * 3, np.eye(3), t, trunc, vol2, [0, 0, 0], col2, rgb, n, True)
    m = col[..., 3] == 1
    assert (col2[..., 3][m] == 2).all() and np.abs(vol2[m].astype(int) - vol[m].astype(int)).max() <= 1
    # raycasting the fused wall returns the wall: z = 1 m in camera frame = 3 m in volume frame, normal along -z (towards the camera)
    vm, nm, cm = np.zeros((3 * rows, cols), np.float32), np.zeros((3 * rows, cols), np.float32), np.zeros((rows, cols, 4), np.uint8)
    S = oracle_mod.raycast(intr, np.eye(3), t, trunc, [size] * 3, vol, vm, nm, [0, 0, 0], cm, col)
    hit = np.isfinite(vm[:rows])
    assert hit[10:38, 10:54].all() and S > 0
    assert np.abs(vm[2 * rows:][hit] - 3.0).max() < 0.5 * cell
    nz = nm[2 * rows:][np.isfinite(nm[:rows])]
    assert (nz < -0.95).all()
    # colour / weight are trilinear over voxels that include never-updated ones behind the band (value 0): 0 < c <= 200
    assert (cm[..., 0][hit] > 0).all() and (cm[..., 0][hit] <= 200).all() and (cm[..., 3][hit] <= 1).all()


def test_icp_identity_and_known_shift(oracle_mod):
    """ICP on identical maps: b == 0 and the solve returns zero motion; a small known translation is recovered."""
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    d, _ = synth.render(synth.Scene("room"), cam, np.eye(3), np.zeros(3))
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    v = oracle_mod.create_vmap(intr, oracle_mod.bilateral_filter(d))
    n = oracle_mod.create_nmap(v)
    t0 = np.array([3, 3, 3], np.float32)
    vg, ng = oracle_mod.transform_maps(v, n, np.eye(3), t0)
    ang = float(np.float32(math.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))
    A, b, r = oracle_mod.icp_step(np.eye(3), t0, v, n, np.eye(3), t0, intr, vg, ng, 0.10, ang, order=0)
    assert r[1] > 0.8 * np.isfinite(n[:120]).sum() and np.abs(b).max() < 1e-3 * np.abs(A).max() and r[0] < 1e-6
    assert np.allclose(A, A.T)
    # perturb the current pose by a known translation: one Gauss-Newton step must point back (x = +delta, see ICPOdometry.cpp:133-178)
    delta = np.array([0.004, -0.003, 0.005], np.float32)
    A, b, r = oracle_mod.icp_step(np.eye(3), t0 + delta, v, n, np.eye(3), t0, intr, vg, ng, 0.10, ang, order=1)
    x = oracle_mod.ldlt_solve6(A.astype(np.float64), b.astype(np.float64))
    assert np.abs(x[:3] + delta).max() < 2e-3 or np.abs(x[:3] - delta).max() < 2e-3
    # float (reference order) and double accumulation agree to float summation error
    Af, bf, _ = oracle_mod.icp_step(np.eye(3), t0 + delta, v, n, np.eye(3), t0, intr, vg, ng, 0.10, ang, order=0)
    assert np.abs(Af - A).max() <= 2e-5 * np.abs(A).max() and np.abs(bf - b).max() <= 2e-5 * np.abs(b).max() + 1e-6


def test_clear_volume_slabs(oracle_mod):
    N = 32
    for axis in range(3):
        vol = np.ones((N, N, N), np.int16)
        oracle_mod.clear_volume(vol, axis, False, 0, 14)  # clearVolume*(wrap, wrap + 14): 15 planes starting at the wrap
        planes = (vol == 0).all(axis=tuple(a for a in range(3) if a != 2 - axis))
        assert planes.sum() == 15 and planes[:15].all()
        vol = np.ones((N, N, N), np.int16)
        oracle_mod.clear_volume(vol, axis, True, 0, -14)  # ...Back: 15 planes ending at the wrap (index 0 and the 14 below it, wrapped)
        planes = (vol == 0).all(axis=tuple(a for a in range(3) if a != 2 - axis))
        assert planes.sum() == 15 and planes[0] and planes[N - 14:].all()
    # quirk A.15: with a 16-voxel shift the X variants launch only 16 x-threads and leave the 17th plane uncleared
    vol = np.ones((N, N, N), np.int16)
    oracle_mod.clear_volume(vol, 0, False, 0, 16)
    assert (vol == 0).all(axis=(0, 1)).sum() == 16
    vol = np.ones((N, N, N), np.int16)
    oracle_mod.clear_volume(vol, 1, False, 0, 16)
    assert (vol == 0).all(axis=(0, 2)).sum() == 17


def test_extract_zero_crossings(oracle_mod):
    """A volume with F = +0.5 for z < 10 and -0.5 for z >= 10 has exactly one +z crossing per (x, y) at the midpoint."""
    N, size = 16, 1.6
    vol = np.zeros((N, N, N), np.int16)
    col = np.zeros((N, N, N, 4), np.uint8)
    vol[:10] = 16383
    vol[10:] = -16383
    col[..., 3] = 7
    col[..., 0], col[..., 1], col[..., 2] = 10, 20, 30
    pts = oracle_mod.extract_cloud_slice(vol, [size] * 3, 10000, [0, 0, 0], col, 0, N, 0, N, 0, N, 1, [0, 0, 0])
    # ... plus the reference's wrap quirk: at z = N - 1 the +z neighbour is fetched modulo N, i.e. plane 0 (extract.cu:190-196),
    # so the sign change between the last and the first plane yields a second, spurious crossing at z = N cells
    assert len(pts) == 2 * N * N
    cell = size / N
    zs = np.sort(pts["xyz"][:, 2])
    assert np.allclose(zs[: N * N], 10 * cell - size / 2, atol=1e-6)  # halfway between voxel centres 9.5 and 10.5 cells
    assert np.allclose(zs[N * N:], N * cell - size / 2, atol=1e-6)
    assert (pts["bgra"] == np.array([10, 20, 30, 7], np.uint8)).all()  # byte order b,g,r,a <- colour.x,y,z and the base weight
    # realVoxelWrap shifts the output by whole cells
    pts2 = oracle_mod.extract_cloud_slice(vol, [size] * 3, 10000, [0, 0, 0], col, 0, N, 0, N, 0, N, 1, [3, 0, -2])
    assert np.allclose(np.sort(pts2["xyz"][:, 0]) - np.sort(pts["xyz"][:, 0]), 3 * cell, atol=1e-6)
    assert np.allclose(np.sort(pts2["xyz"][:, 2]) - np.sort(pts["xyz"][:, 2]), -2 * cell, atol=1e-6)
    # unseen voxels (weight 0) produce nothing
    col[..., 3] = 0
    assert len(oracle_mod.extract_cloud_slice(vol, [size] * 3, 10000, [0, 0, 0], col, 0, N, 0, N, 0, N, 1, [0, 0, 0])) == 0


def test_rgbd_image_kernels(oracle_mod):
    d = np.array([[0, 500, 6000, 6001, 65535]], np.uint16)
    m = oracle_mod.depth_to_metres(d, 6000)
    assert np.isnan(m[0, 0]) and m[0, 1] == np.float32(0.5) and m[0, 2] == np.float32(6.0) and np.isnan(m[0, 3]) and np.isnan(m[0, 4])
    rgb = np.zeros((1, 3, 3), np.uint8)
    rgb[0, 0] = (255, 0, 0)
    rgb[0, 1] = (0, 255, 0)
    rgb[0, 2] = (0, 0, 255)
    i = oracle_mod.bgr_to_intensity(rgb)
    assert list(i[0]) == [int(np.float32(255) * np.float32(0.114)), int(np.float32(255) * np.float32(0.587)), int(np.float32(255) * np.float32(0.299))]
    img = np.full((16, 16), 100, np.uint8)
    assert (oracle_mod.pyr_down_gauss_u8(img) == 100).all()
    dx, dy = oracle_mod.derivative_images(img)
    assert (dx[1:-1, 1:-1] == 0).all() and (dy[1:-1, 1:-1] == 0).all()
    ramp = np.tile(np.arange(16, dtype=np.uint8) * 10, (16, 1))
    dx, dy = oracle_mod.derivative_images(ramp)
    # d/dx of a ramp with step 10: -(2 * 0.52201 + 0.79451) * 20 = -36.77 -> truncated to -36 (kernel is flipped: index 8 first)
    assert (np.abs(dx[2:-2, 2:-2]) == 36).all() and (dy[2:-2, 2:-2] == 0).all()
    f = np.full((16, 16), 2.0, np.float32)
    f[3, 3] = np.nan
    p = oracle_mod.pyr_down_gauss_f32(f)
    assert np.allclose(p, 2.0)  # NaN taps are skipped and the weights renormalised


# ---- -p ground-truth odometry (GroundTruthOdometry.cpp, KintinuousTracker::loadTrajectory) ---------------------------------------
def test_ground_truth_odometry(oracle_mod):
    from kintinuous_amd import synth
    cam = synth.Camera.small(80, 60)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(6)
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    pose7 = synth.ground_truth_rows(traj)
    stamps = np.array([1000 * (k + 1) for k in range(len(traj))], np.uint64)
    cfg = oracle_mod.OTrackerConfig(cam.cols, cam.rows, 32, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
    trk = oracle_mod.OracleTracker(cfg)
    keep = [0, 1, 2, 4, 5]                       # entry 3 is missing from the trajectory: that frame must be dropped (preRun)
    trk.load_trajectory(stamps[keep], pose7[keep])
    C = []
    for R, c in traj:
        C4 = np.eye(4)
        C4[:3, :3], C4[:3, 3] = R, c
        C.append(C4)
    tracked = []
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame(d, rgb, int(stamps[k]))
        if k == 3:
            assert trk.num_poses() == len(tracked), "a frame without a trajectory entry must not be tracked"
            continue
        tracked.append(k)
        assert trk.num_poses() == len(tracked)
        R, t, _ = trk.pose()
        # expected: [I | basis] * C0^-1 Ck, chained through float32 products -> a few ulp
        E = np.eye(4)
        E[:3, 3] = 3.0
        E = E @ np.linalg.inv(C[0]) @ C[k]
        assert np.abs(R - E[:3, :3]).max() < 5e-6 and np.abs(t - E[:3, 3]).max() < 5e-6, k
    # fusion and raycast ran with those poses (a 32^3 volume is coarse: a third of the pixels hit a surface)
    assert np.isfinite(trk.vmap_g_prev(0)[: cam.rows]).mean() > 0.2
    trk.close()

    # quirks: (a) a previous timestamp of 0 reads as "no previous frame" (GroundTruthOdometry.cpp:50) -> frame 1 keeps frame 0's pose;
    # (b) the map's comparator is std::less<int>: timestamps equal modulo 2^32 are the same key
    trk = oracle_mod.OracleTracker(cfg)
    stamps0 = np.array([0, 1000, 2000 + (1 << 32)], np.uint64)
    trk.load_trajectory(stamps0, pose7[:3])
    trk.process_frame(*frames[0], 0)
    R0, t0, _ = trk.pose()
    trk.process_frame(*frames[1], 1000)
    R1, t1, _ = trk.pose()
    assert np.array_equal(R0, R1) and np.array_equal(t0, t1)
    trk.process_frame(*frames[2], 2000)          # found under the narrowed key
    assert trk.num_poses() == 3
    R2, t2, _ = trk.pose()
    E = np.eye(4)
    E[:3, 3] = 3.0
    E = E @ np.linalg.inv(C[1]) @ C[2]           # the motion 0 -> 1 was lost, 1 -> 2 applied
    assert np.abs(R2 - E[:3, :3]).max() < 5e-6 and np.abs(t2 - E[:3, 3]).max() < 5e-6
    trk.close()


def test_view_products(oracle_mod):
    """generateImage / generateDepth on a hand-made map: a plane z = 2 facing the camera, lit from the camera centre."""
    rows, cols = 6, 8
    v = np.zeros((3 * rows, cols), np.float32)
    n = np.zeros((3 * rows, cols), np.float32)
    v[2 * rows:] = 2.0
    n[2 * rows:] = -1.0
    v[0, 0] = np.nan                       # no vertex
    n[1, 1] = np.nan                       # no normal
    col = np.zeros((rows, cols, 4), np.uint8)
    col[..., :3] = (10, 20, 30)
    col[..., 3] = 128                      # heat 1.0 -> pure red (r = 235)
    col[2, 2, 3] = 0                       # heat 0 -> pure blue
    col[3, 3, 3] = 64                      # 0.5 -> value 1.5: between green and yellow -> r = 117 (117.5 truncated), g = 235
    img, dcol = oracle_mod.generate_image(v, n, col, [0.0, 0.0, 0.0])
    assert tuple(img[0, 0]) == (0, 0, 0) and tuple(img[1, 1]) == (0, 0, 0) and tuple(dcol[0, 0]) == (0, 0, 0)
    assert tuple(dcol[4, 4]) == (10, 20, 30)
    # light at the origin, point (0, 0, 2), normal (0, 0, -1): |cos| = 1 -> colour = heat * 1 + 20, stored as (b, g, r)
    assert tuple(img[4, 4]) == (20, 20, 255)
    assert tuple(img[2, 2]) == (255, 20, 20)
    assert tuple(img[3, 3]) == (20, 255, 137)
    d = oracle_mod.generate_depth(np.eye(3, dtype=np.float32), [0.0, 0.0, 0.5], v, n)
    assert d[0, 0] == 0 and d[1, 1] == 0 and d[4, 4] == 1500


def test_dynamic_cube_repositioning(oracle_mod):
    """-d (repositionCube): the cube's corner follows the heading on a circle of half the cube size; it only moves when the camera
    would trip the shift threshold with the new position."""
    from kintinuous_amd import synth
    from scipy.spatial.transform import Rotation
    cam = synth.Camera.small(80, 60)
    scene = synth.Scene("room")
    cfg = oracle_mod.OTrackerConfig(cam.cols, cam.rows, 64, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 3, 2, 0, 0, 0, 0, 0, 0, 1)
    trk = oracle_mod.OracleTracker(cfg)
    assert np.allclose(trk.volume_basis(), [3.0, 3.0, 0.0])          # KintinuousTracker.cpp:103-106 without the -sm offset
    yaws = [0.02 * k for k in range(10)]
    moved_at = None
    for k, a in enumerate(yaws):
        R = Rotation.from_euler("y", a).as_matrix()
        d, rgb = synth.render(scene, cam, R, np.zeros(3))
        trk.process_frame(d, rgb, 33333 * k)
        b = trk.volume_basis()
        Rt, t, _ = trk.pose()
        if moved_at is None and not np.allclose(b, [3.0, 3.0, 0.0]):
            moved_at = k
            heading = Rotation.from_matrix(Rt.astype(np.float64)).as_rotvec()[1]
            # the new corner: radius * (cos(heading + pi / 2) + 1), radius * (sin(heading - pi / 2) + 1)
            assert abs(b[0] - 3.0 * (1 - np.sin(heading))) < 1e-4 and abs(b[2] - 3.0 * (1 - np.cos(heading))) < 1e-4 and b[1] == 3.0
            # it moved because the camera was then >= 3 voxels from it, which also shifted the volume in the same frame
            assert np.abs(trk.voxel_wrap()).max() >= 3
    assert moved_at is not None and moved_at >= 2                    # not before the heading has swung the corner 3 voxels away
    assert trk.num_poses() == len(yaws)
    trk.close()


def test_place_recognition_tap_known_answer(oracle_mod):
    """KintinuousTracker.cpp:601-624, 706-717, 917-958 on -p ground truth (exact poses): a camera translating 40 mm per frame along x
    passes the movement threshold ((|rotation| + |translation|) / 2 >= 0.15, i.e. 0.30 m without rotation) every 8th frame -- unless
    a volume shift comes first and takes the pending sample with it (shiftSend).  Frame 0 and the final slice are always sampled."""
    from kintinuous_amd import synth
    from oracle.oracle import OTrackerConfig, OracleTracker
    cam = synth.Camera.small(64, 48)
    scene = synth.Scene("wall")
    n = 24
    poses = [(np.eye(3), np.array([0.04 * k, 0.0, 0.0])) for k in range(n)]
    frames = [synth.render(scene, cam, R, c) for R, c in poses]
    stamps = np.array([1000 * (k + 1) for k in range(n)], np.uint64)
    rows = synth.ground_truth_rows(poses)
    # 7 m / 32 voxels = 0.21875 m per voxel; threshold 2 voxels = 0.4375 m: shifts at frames 11 (0.44 m) and 22
    trk = OracleTracker(OTrackerConfig(cam.cols, cam.rows, 32, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 2, 2, 0, 0, 0, 0, 0, 0, 0, 1))
    trk.load_trajectory(stamps, rows)
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame(d, rgb, int(stamps[k]))
    trk.finalise()
    loops = [k for k in range(trk.num_poses()) if trk.dense_pose(k)[2]]
    # 0.32 m reached at frame 8 -> sample; from there 0.32 m more at frame 16 -> sample; the shift at frame 11 happens with a sample
    # pending (shiftSend), so it samples too and restarts the distance: 8, 11, 19, (shift at 22: sample), ...
    assert loops == [0, 8, 11, 19, 22], loops
    samples = trk.pr_samples()
    assert [int(s[0]) for s in samples] == [1000, 9000, 12000, 20000, 23000, 24000]      # + the final slice's sample (last frame's stamp)
    assert np.allclose(samples[1][1], [0.32, 0, 0], atol=1e-5) and np.allclose(samples[1][2], np.eye(3))
    assert trk.num_slices() == 3
    assert [trk.slice_pr_id(i) for i in range(3)] == [2, 4, 5]     # the two shift slices and the final one carry their samples
    trk.close()
    # without a vocabulary nothing is sampled and only frame 0 is a loop pose
    trk = OracleTracker(OTrackerConfig(cam.cols, cam.rows, 32, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 2, 2, 0, 0, 0, 0, 0, 0, 0, 0))
    trk.load_trajectory(stamps, rows)
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame(d, rgb, int(stamps[k]))
    assert [k for k in range(trk.num_poses()) if trk.dense_pose(k)[2]] == [0] and trk.pr_samples() == []
    trk.close()


def test_slice_stage_trig_restatement_is_pinned_to_libm(oracle_mod):
    """Advisor, round 4: HIP (csrc/kt_slice.hip) and the oracle share ONE hand-written atan2 / sin / cos (Cephes kernels, every FMA written
    out) so that the per-slice normals compare bit for bit -- which proves only that both sides run the same code.  This pins the shared
    restatement itself: against libm's double-precision atan2 / sin / cos it must stay within a few units in the last place on the whole
    domain the stage uses (atan2(y >= 0, x) in [0, pi]; sin / cos on [0, pi / 3]), so a wrong coefficient or range reduction fails here."""
    import ctypes as C
    import math
    l = oracle_mod.lib()
    f = l.kto_test_sp_atan2_pos
    f.restype, f.argtypes = C.c_float, [C.c_float, C.c_float]
    g = l.kto_test_sp_sincos
    g.restype, g.argtypes = None, [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(7)
    ys = (np.abs(rng.normal(size=40000)) * 10.0 ** rng.uniform(-6, 3, 40000)).astype(np.float32)
    xs = (rng.normal(size=40000) * 10.0 ** rng.uniform(-6, 3, 40000)).astype(np.float32)
    ys[:200], xs[200:400] = 0.0, 0.0                            # the axes; (0, 0) is defined as 0 below
    worst = 0.0
    for y, x in zip(ys.tolist(), xs.tolist()):
        r, t = f(y, x), math.atan2(y, x)
        assert 0.0 <= r <= 3.1415928
        worst = max(worst, abs(r - t) / float(np.spacing(np.float32(max(t, 1e-30)))))
    assert worst <= 4.0, worst                                   # measured 3.03
    assert f(0.0, 0.0) == 0.0 and abs(f(0.0, -1.0) - math.pi) < 3e-7 and f(0.0, 2.0) == 0.0 and abs(f(3.0, 0.0) - math.pi / 2) < 2e-7
    s, c = C.c_float(), C.c_float()
    ws = wc = 0.0
    for t in np.linspace(0.0, math.pi / 3 + 1e-6, 40001).astype(np.float32).tolist():
        g(t, C.byref(s), C.byref(c))
        ws = max(ws, abs(s.value - math.sin(t)) / float(np.spacing(np.float32(max(math.sin(t), 1e-30)))))
        wc = max(wc, abs(c.value - math.cos(t)) / float(np.spacing(np.float32(math.cos(t)))))
    assert ws <= 2.5 and wc <= 2.5, (ws, wc)                     # measured 1.75 / 1.56
