"""GPU parity AT THE CONFIGURATIONS BASELINE.json QUOTES: the HIP tracker against the oracle tracker on the
full-size inputs the bench lines are measured on -- not reduced stand-ins.
  config 2: 640x480 orbit, ICP only, 512^3, through the first volume shift (frames are pushed with read-ahead, as bench.py does);
  config 3: 640x480 crab-walk, `-s 7 -ri` (ICP + RGB-D), 512^3, default -t 14, across the first two X shifts;
  config 5: 1280x960 far wall, 768^3, static mode (`-sm`), the roofline stress case;
  the reductions at 1280x960 level 0 (more than KT_KBATCH k-steps per virtual thread: the multi-batch loop of kt_reduce29);
  a long shifting sequence at reduced size (the former tests/tools/soak.py), every pose compared.
Bars: shift decisions and voxel wraps identical, per-frame pose within 1e-6 (the only non-bit-exact piece is the device's
double-precision sin / cos in Rodrigues), TSDF + colour/weight volumes byte-identical, slices identical as point sets."""
import os

import numpy as np
import pytest

from test_gpu_tracker import _cfgs, _same_points, _volume_close

pytestmark = pytest.mark.gpu


def _push_all(trk, otr, ctx, frames, readahead=True, check_every=1):
    """HIP: device-resident frames with one frame of read-ahead (bench.py's loop); oracle: frame by frame.  Poses are compared
    from the dense pose graphs afterwards, so the HIP side is never drained between frames (getters after every frame
    hide missing cross-stream ordering)."""
    dev = [(ctx.upload(d), ctx.upload(c)) for d, c in frames]
    for k in range(len(frames)):
        if readahead and k + 1 < len(frames):
            trk.prefetch_frame(*dev[k + 1])
        trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
    for k, (d, c) in enumerate(frames):
        otr.process_frame(d, c, 33333 * k)
    assert trk.num_poses() == otr.num_poses() == len(frames)
    worst = 0.0
    for i in range(len(frames)):
        ts, p, _ = trk.dense_pose(i)
        ots, op, _ = otr.dense_pose(i)
        assert ts == ots
        worst = max(worst, float(np.abs(p - op).max()))
    return worst


def _same_slices(trk, otr):
    assert trk.num_slices() == otr.num_slices()
    for i in range(trk.num_slices()):
        p, dim = trk.slice(i)
        q, odim = otr.slice(i)
        assert dim == odim and _same_points(p, q), f"slice {i}: dimension {dim}/{odim}, {len(p)}/{len(q)} points"


def test_config2_orbit512_through_the_first_shift(ctx, oracle_mod):
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = synth.Camera()
    _, frames, traj, kw = synth.sequence("orbit", 34, cam)
    g, o = _cfgs(cam, 512, **kw)
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    worst = _push_all(trk, otr, ctx, frames)
    assert worst < 1e-6, worst
    assert np.array_equal(trk.voxel_wrap(), otr.voxel_wrap()) and np.abs(otr.voxel_wrap()).sum() > 0, "the sequence must include a shift"
    _same_slices(trk, otr)
    assert trk.num_slices() >= 1
    _volume_close(trk, otr)
    R, t, gc = trk.pose()
    assert np.abs(gc - (traj[-1][1])).max() < 0.01          # and it tracks the ground truth
    trk.close(); otr.close()


def test_config3_crabwalk512_rgbd_icp_two_shifts(ctx, oracle_mod):
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = synth.Camera()
    _, frames, traj, kw = synth.sequence("crabwalk", 29, cam)
    assert kw["volume_size"] == 7.0 and kw["use_rgbd_icp"] == 1
    g, o = _cfgs(cam, 512, **kw)
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    worst = _push_all(trk, otr, ctx, frames)
    assert worst < 1e-6, worst
    assert np.array_equal(trk.voxel_wrap(), otr.voxel_wrap())
    assert otr.voxel_wrap()[0] >= 28 and otr.num_slices() >= 2, (otr.voxel_wrap(), otr.num_slices())
    _same_slices(trk, otr)
    _volume_close(trk, otr)
    trk.close(); otr.close()


def test_config5_farwall768_static(ctx, oracle_mod):
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = synth.Camera(1280, 960, 2 * synth.FX, 2 * synth.FY, 2 * synth.CX, 2 * synth.CY)
    _, frames, traj, kw = synth.sequence("farwall", 2, cam)
    g, o = _cfgs(cam, 768, **kw)
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    worst = _push_all(trk, otr, ctx, frames)
    assert worst < 1e-6, worst
    U = otr.last_counts()[0]
    assert U > 1e8, U                                       # the stress case: > 10^8 voxels updated per frame
    v, ov = trk.volume(), otr.volume()
    assert np.array_equal(v, ov), int((v != ov).sum())
    del v, ov
    c, oc = trk.color_volume(), otr.color_volume()
    assert np.array_equal(c, oc), int((c != oc).any(axis=-1).sum())
    del c, oc
    for lvl in range(4):
        a, b = trk.vmap_g_prev(lvl), otr.vmap_g_prev(lvl)
        rows = a.shape[0] // 3
        va = np.isfinite(a[:rows])
        assert np.array_equal(va, np.isfinite(b[:rows])) and va.sum() > 0
        assert np.array_equal(a[:rows][va].view(np.uint32), b[:rows][va].view(np.uint32))
    trk.close(); otr.close()


def test_reductions_at_1280x960(ctx, oracle_mod):
    """kt_icp_step / kt_rgb_residual / kt_rgb_step at 1280x960 level 0: 150 pixels per virtual thread > KT_KBATCH = 40, so the
    staged reduction runs its batch loop 4 times."""
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
    cam = synth.Camera(1280, 960, 2 * synth.FX, 2 * synth.FY, 2 * synth.CX, 2 * synth.CY)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(300)
    (d0, rgb0), (d1, rgb1) = [synth.render(scene, cam, *traj[i]) for i in (0, 1)]
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    v0 = O.create_vmap(intr, O.bilateral_filter(d0)); n0 = O.create_nmap(v0)
    v1 = O.create_vmap(intr, O.bilateral_filter(d1)); n1 = O.create_nmap(v1)
    R0, t0 = np.asarray(traj[0][0], np.float32), (np.asarray(traj[0][1], np.float32) + 3).astype(np.float32)
    vg, ng = O.transform_maps(v0, n0, R0, t0)
    ang = float(np.sin(np.float32(20.0 * 3.14159265 / 180.0)))
    Ao, bo, ro = O.icp_step(R0, t0, v1, n1, O.mat33_inverse(R0), t0, intr, vg, ng, 0.10, ang, 0)
    A, b, r = H.icp_step(R0, t0, v1, n1, O.mat33_inverse(R0), t0, intr, vg, ng, 0.10, ang)
    assert ro[1] > 5e5
    assert np.array_equal(A.view(np.uint32), Ao.view(np.uint32)) and np.array_equal(b.view(np.uint32), bo.view(np.uint32))
    assert np.array_equal(np.asarray(r, np.float32).view(np.uint32), ro.view(np.uint32))
    dm0, dm1 = O.depth_to_metres(d0, 6000), O.depth_to_metres(d1, 6000)
    i0, i1 = O.bgr_to_intensity(rgb0), O.bgr_to_intensity(rgb1)
    dx, dy = O.derivative_images(i1)
    K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]])
    krk = (K @ O.rodrigues(np.array([0.001, -0.002, 0.0005])) @ np.linalg.inv(K)).astype(np.float32)
    kt = (K @ np.array([0.002, -0.001, 0.001])).astype(np.float32)
    ms = float(np.float32(12.0 ** 2 / 0.125 ** 2))
    co, so, no = O.rgb_residual(ms, dx, dy, dm0, dm1, i0, i1, np.float32(0.07), kt, krk)
    ch, sh, nh = H.rgb_residual(ms, dx, dy, dm0, dm1, i0, i1, np.float32(0.07), kt, krk)
    assert no > 200 and (sh, nh) == (so, no)
    m = co["valid"] != 0
    assert np.array_equal(ch["valid"] != 0, m)
    for f in ("zero", "one"):
        assert np.array_equal(ch[f][m], co[f][m])
    assert np.array_equal(ch["diff"][m].view(np.uint32), co["diff"][m].view(np.uint32))
    cloud = O.project_to_cloud(dm0, cam.fx, cam.fy, cam.cx, cam.cy, 0)
    sig = float(np.sqrt(np.float32(no)))
    Ao, bo = O.rgb_step(co, sig, cloud, np.float32(cam.fx), np.float32(cam.fy), dx, dy, 0.125, 0)
    A, b = H.rgb_step(co, sig, cloud, np.float32(cam.fx), np.float32(cam.fy), dx, dy, 0.125)
    assert np.array_equal(A.view(np.uint32), Ao.view(np.uint32)) and np.array_equal(b.view(np.uint32), bo.view(np.uint32))


@pytest.mark.parametrize("mode", ["icp", "rgbd_icp"])
def test_soak_shifting_sequence(ctx, oracle_mod, mode):
    """Long shifting crab-walk at 160x120 / 96^3 with a 3-voxel shift threshold (a shift every few frames, X+ then X-): every pose,
    every shift decision, every slice and the final volumes against the oracle."""
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    n = 260 if mode == "icp" else 90
    frames = [synth.render(scene, cam, *traj[i]) for i in range(n)]
    g, o = _cfgs(cam, 96, volume_size=7.0, voxel_shift=3, use_rgbd_icp=int(mode == "rgbd_icp"))
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    worst = _push_all(trk, otr, ctx, frames)
    assert worst < 1e-5, worst
    assert np.array_equal(trk.voxel_wrap(), otr.voxel_wrap())
    assert otr.num_slices() >= (10 if mode == "icp" else 5)
    _same_slices(trk, otr)
    _volume_close(trk, otr)
    trk.close(); otr.close()


def test_ground_truth_mode_with_a_lagging_gpu(ctx, oracle_mod):
    """-p (poses from a trajectory file): the host never waits for the device, so with device-resident frames it runs frames ahead
    of a GPU that needs ~1.5 ms per 1280x960 / 512^3 fusion.  The read-ahead stream recycles frame sets behind odo_ev only
    (without explicit ordering a recycled set was overwritten while an earlier frame's fusion was still queued).
    All frames are pushed back to back, nothing is read until the end."""
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = synth.Camera(1280, 960, 2 * synth.FX, 2 * synth.FY, 2 * synth.CX, 2 * synth.CY)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(300)
    poses = [traj[i] for i in range(0, 24, 2)]
    frames = [tuple(np.ascontiguousarray(a) for a in synth.render(scene, cam, *p)) for p in poses]
    stamps = np.array([33333 * (k + 1) for k in range(len(frames))], np.uint64)
    rows = synth.ground_truth_rows(poses)
    g, o = _cfgs(cam, 512)
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    trk.load_trajectory(stamps, rows)
    otr.load_trajectory(stamps, rows)
    dev = [(ctx.upload(d), ctx.upload(c)) for d, c in frames]
    ctx.sync()
    for k in range(len(frames)):
        if k + 1 < len(frames):
            trk.prefetch_frame(*dev[k + 1])
        trk.process_frame(dev[k][0], dev[k][1], int(stamps[k]))
    for k, (d, c) in enumerate(frames):
        otr.process_frame(d, c, int(stamps[k]))
    assert trk.num_poses() == otr.num_poses() == len(frames)
    for i in range(len(frames)):
        a, b = trk.dense_pose(i), otr.dense_pose(i)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
    _volume_close(trk, otr)
    trk.close(); otr.close()


def test_integrate_pointer_path(ctx, oracle_mod, small_scene, monkeypatch):
    """tsdf23's pointer-addressed variant (kt_tsdf23_kernel<*, false>: volumes beyond the 32-bit byte offsets of a buffer descriptor,
    N >= 1024) forced at a small N: accumulated frames, wrapped storage, against the oracle."""
    from hip_kernels import HipKernels
    from oracle.oracle import OIntr
    monkeypatch.setenv("KT_TSDF_POINTERS", "1")
    cam, frames, traj = small_scene
    O, H = oracle_mod, HipKernels(ctx)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    N, size, wrap = 96, 6.0, [5, 90, 41]
    trunc = max(0.06, 2.1 * size / N)
    vo, co = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    vh, ch = vo.copy(), co.copy()
    for k in range(3):
        d, c = frames[k]
        n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
        Rk = np.asarray(traj[k][0], np.float32)
        tk = (np.asarray(traj[k][1], np.float32) + 3).astype(np.float32)
        Rinv = O.mat33_inverse(Rk)
        O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, True)
        H.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vh, wrap, ch, c, n, True)
        assert np.array_equal(vo, vh) and np.array_equal(co, ch), k


def test_integrate_1024_cubed(ctx, oracle_mod):
    """The same variant where it is needed: one 640x480 frame into a 1024^3 volume (2 GiB of tsdf, 4 GiB of colour), voxel indices past
    2^30, against the oracle.  Needs ~14 GB of host memory for the two copies."""
    import psutil
    if psutil.virtual_memory().available < 24e9:
        pytest.skip("not enough host memory for two 1024^3 volume pairs")
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
    cam = synth.Camera()
    d, c = synth.render(synth.Scene("room"), cam, *synth.orbit_trajectory(2)[0])
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
    N, size, wrap = 1024, 6.0, [1000, 3, 517]
    trunc = max(0.06, 2.1 * size / N)
    vo, co = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    U, _ = O.integrate_tsdf(d, intr, [size] * 3, np.eye(3), [3, 3, 3], trunc, vo, wrap, co, c, n, True)
    assert U > 2e7
    dv, dc = ctx.zeros(N ** 3 * 2), ctx.zeros(N ** 3 * 4)
    sc = ctx.zeros(cam.rows * cam.cols * 4)
    ctx.integrate_tsdf(ctx.upload(d), cam.cols, cam.rows, H._intr(intr), [size] * 3, np.eye(3), [3, 3, 3], trunc, dv, sc, wrap, dc, ctx.upload(c),
                       ctx.upload(n), True, N)
    ctx.sync()
    vh = ctx.download(dv, np.int16, (N, N, N))
    assert np.array_equal(vo, vh)
    del vh, vo
    chh = ctx.download(dc, np.uint8, (N, N, N, 4))
    assert np.array_equal(co, chh)


def test_integrate_without_the_depth_range_prune(ctx, oracle_mod):
    """An image with more than 8192 pixel tiles (here 4096x2176: 128 x 68 tiles of 32 x 32) does not fit the interval pre-pass's tile-max
    table, so the depth-range prune is switched off and only the frustum bounds the column intervals.  The results may not depend on how
    tight the intervals are: two frames into a 128^3 volume, against the oracle."""
    from hip_kernels import HipKernels
    from kintinuous_amd import synth
    from oracle.oracle import OIntr
    O, H = oracle_mod, HipKernels(ctx)
    cols, rows = 4096, 2176
    cam = synth.Camera.small(cols, rows)
    assert ((cols + 31) // 32) * ((rows + 31) // 32) > 8192
    poses = synth.orbit_trajectory(6)
    intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    N, size, wrap = 128, 6.0, [3, 120, 64]
    trunc = max(0.06, 2.1 * size / N)
    vo, co = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    vh, ch = vo.copy(), co.copy()
    for k in (0, 5):
        d, c = synth.render(synth.Scene("room"), cam, *poses[k])
        n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
        Rk = np.asarray(poses[k][0], np.float32)
        tk = (np.asarray(poses[k][1], np.float32) + 3).astype(np.float32)
        Rinv = O.mat33_inverse(Rk)
        O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, True)
        H.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vh, wrap, ch, c, n, True)
        assert np.array_equal(vo, vh) and np.array_equal(co, ch), k
    assert int((co[..., 3] != 0).sum()) > 30000
