"""GPU parity: tracking reductions through the C-ABI vs the oracle.
Per-pixel work is bit-identical and the 29 float sums are folded in the reference's exact order (64 x 128 grid-stride
partials, 32-lane shuffle trees, reduceSum<<<1,512>>>), so A, b, residual must equal the oracle's reference-order result
BIT FOR BIT; the oracle's double-precision accumulation bounds the float summation error of both.
Integer outputs (inlier count, DataTerm image, count / sigma) are exact."""
import numpy as np
import pytest

from conftest import random_rotation

pytestmark = pytest.mark.gpu


def _frame_maps(oracle, cam, depth, level):
    from oracle.oracle import OIntr
    d = oracle.bilateral_filter(depth)
    for _ in range(level):
        d = oracle.pyr_down(d)
    v = oracle.create_vmap(OIntr(cam.fx, cam.fy, cam.cx, cam.cy).level(level), d)
    return v, oracle.create_nmap(v)


def _close(hip, ref_f, ref_d, name):
    scale = np.abs(ref_d).max() + 1e-12
    # the reference-order float result and the HIP result must both sit within float summation error of the double sum
    assert np.abs(hip - ref_d).max() <= 2e-5 * scale, f"{name}: hip vs double {np.abs(hip - ref_d).max() / scale:.3e}"
    assert np.abs(ref_f - ref_d).max() <= 2e-5 * scale, f"{name}: oracle float vs double {np.abs(ref_f - ref_d).max() / scale:.3e}"


@pytest.mark.parametrize("level", [0, 1, 2])
def test_icp_step(ctx, oracle_mod, small_scene, level):
    from kintinuous_amd.abi import Intr
    from oracle.oracle import OIntr
    cam, frames, traj = small_scene
    rng = np.random.default_rng(level)
    vc, nc = _frame_maps(oracle_mod, cam, frames[1][0], level)
    v0, n0 = _frame_maps(oracle_mod, cam, frames[0][0], level)
    t0 = np.array([3, 3, 3], np.float32)
    vg, ng = oracle_mod.transform_maps(v0, n0, np.eye(3), t0)
    rows, cols = vc.shape[0] // 3, vc.shape[1]
    Rprev = random_rotation(rng, 0.02)
    Rprev_inv = oracle_mod.mat33_inverse(Rprev)
    Rcurr = (random_rotation(rng, 0.01) @ Rprev).astype(np.float32)
    tcurr = t0 + rng.uniform(-0.01, 0.01, 3).astype(np.float32)
    dist, ang = 0.10, float(np.float32(np.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))
    oi, gi = OIntr(cam.fx, cam.fy, cam.cx, cam.cy).level(level), Intr(cam.fx, cam.fy, cam.cx, cam.cy).level(level)
    Af, bf, rf = oracle_mod.icp_step(Rcurr, tcurr, vc, nc, Rprev_inv, t0, oi, vg, ng, dist, ang, order=0)
    Ad, bd, rd = oracle_mod.icp_step(Rcurr, tcurr, vc, nc, Rprev_inv, t0, oi, vg, ng, dist, ang, order=1)
    A, b, r = ctx.icp_step(Rcurr, tcurr, ctx.upload(vc), ctx.upload(nc), Rprev_inv, t0, gi, ctx.upload(vg), ctx.upload(ng), cols, rows, dist, ang)
    assert rd[1] > 0.3 * rows * cols
    assert r[1] == rd[1] == rf[1], "inlier count must be exact"
    _close(A, Af, Ad, "A")
    _close(b, bf, bd, "b")
    assert abs(r[0] - rd[0]) <= 2e-5 * abs(rd[0])
    assert np.array_equal(A.view(np.uint32), Af.view(np.uint32)), "A must match the reference-order float sums bit for bit"
    assert np.array_equal(b.view(np.uint32), bf.view(np.uint32)) and np.array_equal(r.view(np.uint32), rf.view(np.uint32))
    # run-to-run determinism of the fixed-order fold
    A2, b2, r2 = ctx.icp_step(Rcurr, tcurr, ctx.upload(vc), ctx.upload(nc), Rprev_inv, t0, gi, ctx.upload(vg), ctx.upload(ng), cols, rows, dist, ang)
    assert np.array_equal(A, A2) and np.array_equal(b, b2) and np.array_equal(r, r2)


def test_icp_step_all_invalid(ctx, oracle_mod, small_scene):
    from kintinuous_amd.abi import Intr
    cam, _, _ = small_scene
    rows, cols = cam.rows, cam.cols
    nanmap = np.full((3 * rows, cols), np.nan, np.float32)
    A, b, r = ctx.icp_step(np.eye(3), [3, 3, 3], ctx.upload(nanmap), ctx.upload(nanmap), np.eye(3), [3, 3, 3], Intr(cam.fx, cam.fy, cam.cx, cam.cy),
                           ctx.upload(nanmap), ctx.upload(nanmap), cols, rows, 0.1, 0.34)
    assert not A.any() and not b.any() and r[1] == 0


@pytest.mark.parametrize("level", [0, 1])
def test_rgb_residual_and_step(ctx, oracle_mod, small_scene, level):
    from kintinuous_amd.abi import DATATERM_DTYPE
    cam, frames, traj = small_scene

    def pyr(depth, rgb):
        dm, im = oracle_mod.depth_to_metres(depth, 6000), oracle_mod.bgr_to_intensity(rgb)
        for _ in range(level):
            dm, im = oracle_mod.pyr_down_gauss_f32(dm), oracle_mod.pyr_down_gauss_u8(im)
        return dm, im

    ld, li = pyr(*frames[0])
    nd, ni = pyr(*frames[1])
    rows, cols = ni.shape
    dx, dy = oracle_mod.derivative_images(ni)
    div = 1 << level
    K = np.array([[cam.fx / div, 0, cam.cx / div], [0, cam.fy / div, cam.cy / div], [0, 0, 1]])
    Rinc = oracle_mod.rodrigues([0.002, -0.003, 0.001])
    krk = (K @ Rinc @ np.linalg.inv(K)).astype(np.float32)
    kt = (K @ np.array([0.004, -0.002, 0.003])).astype(np.float32)
    min_scale = float(np.float32((12.0 if level == 0 else 5.0) ** 2 / 0.125 ** 2))
    corres, sigma, count = oracle_mod.rgb_residual(min_scale, dx, dy, ld, nd, li, ni, np.float32(0.07), kt, krk)
    gcor = ctx.zeros(rows * cols * 16)
    gdx, gdy = ctx.upload(dx), ctx.upload(dy)
    gs, gc = ctx.rgb_residual(min_scale, gdx, gdy, ctx.upload(ld), ctx.upload(nd), ctx.upload(li), ctx.upload(ni), cols, rows, gcor,
                              float(np.float32(0.07)), kt, krk)
    assert count > 50
    assert (gs, gc) == (sigma, count)
    got = ctx.download(gcor, DATATERM_DTYPE, (rows, cols))
    assert np.array_equal(got["valid"], corres["valid"])
    m = corres["valid"] == 1
    for f in ("zero", "one"):
        assert np.array_equal(got[f][m], corres[f][m])
    assert np.array_equal(got["diff"][m].view(np.uint32), corres["diff"][m].view(np.uint32))
    # rgbStep on the oracle's correspondence image
    cloud = oracle_mod.project_to_cloud(ld, cam.fx, cam.fy, cam.cx, cam.cy, level)
    sig_val = float(np.sqrt(np.float32(count)))
    fx, fy = np.float32(cam.fx) / np.float32(div), np.float32(cam.fy) / np.float32(div)
    Af, bf = oracle_mod.rgb_step(corres, sig_val, cloud, fx, fy, dx, dy, 0.125, order=0)
    Ad, bd = oracle_mod.rgb_step(corres, sig_val, cloud, fx, fy, dx, dy, 0.125, order=1)
    A, b = ctx.rgb_step(ctx.upload(corres), sig_val, ctx.upload(cloud), float(fx), float(fy), gdx, gdy, 0.125, cols, rows)
    scale = np.abs(Ad).max()
    assert np.abs(A - Ad).max() <= 2e-5 * scale and np.abs(Af - Ad).max() <= 2e-5 * scale
    assert np.array_equal(A.view(np.uint32), Af.view(np.uint32)) and np.array_equal(b.view(np.uint32), bf.view(np.uint32))


def test_conversion_semantics(ctx, oracle_mod):
    """CUDA __float2int_r{n,z,d} saturate and map NaN to 0; gfx950's v_cvt_i32_f32 must agree with the oracle's
    explicit emulation.  Exercised through createVMap-independent paths is awkward, so probe via bilateral + pyrDown
    on crafted inputs (rn ties, truncation) and via raycast's floor (covered in test_gpu_volume)."""
    # rn ties: a constant image filtered by the bilateral kernel returns itself; half-way values arise in pyrDown truncation
    d = np.full((64, 64), 1001, np.uint16)
    d[::2, ::2] = 1000
    ref = oracle_mod.pyr_down(oracle_mod.bilateral_filter(d))
    g0, g1 = ctx.empty(d.nbytes), ctx.empty(d.nbytes // 4)
    ctx.bilateral_filter(ctx.upload(d), g0, 64, 64)
    ctx.pyr_down(g0, 64, 64, g1)
    assert np.array_equal(ref, ctx.download(g1, np.uint16, ref.shape))


@pytest.mark.parametrize("schedule", [(10, 5, 4, 0), (0, 10, 5, 0), (2, 0, 0, 3), (0, 0, 0, 0)])
def test_icp_track_matches_the_stepwise_loop(ctx, oracle_mod, small_scene, schedule):
    """kt_icp_track (SURVEY 8(b) export list: the fused multi-iteration odometry, device-side solve, pose in / pose out) against the
    loop the reference runs on the host -- kt_icp_step, kt_host_ldlt_solve6, kt_host_pose_update per iteration (ICPOdometry.cpp:88-180):
    the final pose, the last iteration's A and residual must be identical bit for bit."""
    from kintinuous_amd import abi
    from kintinuous_amd.abi import Intr
    cam, frames, traj = small_scene
    t0 = np.array([3, 3, 3], np.float32)
    Rprev = random_rotation(np.random.default_rng(5), 0.02)
    cur, prev = [], []
    for l in range(4):
        vc, nc = _frame_maps(oracle_mod, cam, frames[1][0], l)
        v0, n0 = _frame_maps(oracle_mod, cam, frames[0][0], l)
        vg, ng = oracle_mod.transform_maps(v0, n0, Rprev, t0)
        cur.append((ctx.upload(vc), ctx.upload(nc)))
        prev.append((ctx.upload(vg), ctx.upload(ng)))
    dist, ang = 0.10, float(np.float32(np.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))
    gi = Intr(cam.fx, cam.fy, cam.cx, cam.cy)
    Rc, tc, A, r = ctx.icp_track([c[0] for c in cur], [c[1] for c in cur], [p[0] for p in prev], [p[1] for p in prev], cam.cols, cam.rows, gi, Rprev, t0,
                                 schedule, dist, ang)
    # the same iterations one at a time, with the solve on the host
    R, t = Rprev.copy(), t0.copy()
    Rprev_inv = abi.host_mat33_inverse(Rprev)
    rt = np.eye(4)
    lastA, lastr = np.zeros((6, 6), np.float32), np.zeros(2, np.float32)
    for l in (3, 2, 1, 0):
        for _ in range(schedule[l]):
            lastA, b, lastr = ctx.icp_step(R, t, cur[l][0], cur[l][1], Rprev_inv, t0, gi.level(l), prev[l][0], prev[l][1], cam.cols >> l, cam.rows >> l, dist, ang)
            x = abi.host_ldlt_solve6(lastA.astype(np.float64), b.astype(np.float64))
            rt, R, t = abi.host_pose_update(x, rt, Rprev, t0)
    assert np.array_equal(Rc.view(np.uint32), np.asarray(R, np.float32).view(np.uint32)) and np.array_equal(tc.view(np.uint32), np.asarray(t, np.float32).view(np.uint32))
    assert np.array_equal(A.view(np.uint32), lastA.view(np.uint32)) and np.array_equal(r.view(np.uint32), lastr.view(np.uint32))
    if sum(schedule):
        assert np.abs(tc - t0).max() > 1e-5 and r[1] > 1000    # it did track something


def _handoff_fault(ctx, skip, count, spin_limit=0, want_dirty=True):
    import ctypes as C
    from kintinuous_amd import abi
    dirty = C.c_uint(0)
    abi._chk(abi.lib().kt_debug_handoff_fault(ctx.h, skip, count, spin_limit, C.byref(dirty) if want_dirty else None))
    return dirty.value


def test_a_reported_handoff_timeout_leaves_a_clean_buffer(ctx, oracle_mod, small_scene):
    """Advisor, round 4: the sentinel hand-off ("the data is the flag") has no tag to tell a stale granule from a fresh one, so a sweep
    that gives up -- it hands nothing back, and the publishers it did not wait for store after it has looked -- must not leave the buffer
    to the next launch as it is.  kt_debug_handoff_fault makes ONE reduction launch lose a publishing workgroup (its sweep gives up after
    a lowered number of looks).  The call must report the time-out, the buffer must be all sentinels again once the stream has drained
    (every reporting path refills it: kt_refill_granules), and the next calls must give the reference-order sums bit for bit."""
    from kintinuous_amd import abi
    from kintinuous_amd.abi import Intr
    from oracle.oracle import OIntr
    cam, frames, traj = small_scene
    vc, nc = _frame_maps(oracle_mod, cam, frames[1][0], 0)
    v0, n0 = _frame_maps(oracle_mod, cam, frames[0][0], 0)
    t0 = np.array([3, 3, 3], np.float32)
    vg, ng = oracle_mod.transform_maps(v0, n0, np.eye(3), t0)
    rows, cols = vc.shape[0] // 3, vc.shape[1]
    R = np.eye(3, dtype=np.float32)
    dist, ang = 0.10, 0.342
    oi, gi = OIntr(cam.fx, cam.fy, cam.cx, cam.cy), Intr(cam.fx, cam.fy, cam.cx, cam.cy)
    Af, bf, rf = oracle_mod.icp_step(R, t0, vc, nc, R, t0, oi, vg, ng, dist, ang, order=0)
    dev = [ctx.upload(x) for x in (vc, nc, vg, ng)]
    step = lambda: ctx.icp_step(R, t0, dev[0], dev[1], R, t0, gi, dev[2], dev[3], cols, rows, dist, ang)
    A, b, r = step()
    assert np.array_equal(A, Af) and np.array_equal(b, bf) and np.array_equal(r, rf)
    assert _handoff_fault(ctx, 0, 0) == 0                                  # a completed sweep hands every granule back
    try:
        _handoff_fault(ctx, 0, 1, spin_limit=64, want_dirty=False)
        with pytest.raises(abi.KtError, match="timed out"):
            step()
        assert _handoff_fault(ctx, 0, 0) == 0, "a reported time-out must refill the hand-off buffer"
        for _ in range(3):
            A, b, r = step()
            assert np.array_equal(A, Af) and np.array_equal(b, bf) and np.array_equal(r, rf)
        # the same through the device-resident chain (kt_icp_track): the LAST of its launches loses the publisher
        pyr = lambda depth: [_frame_maps(oracle_mod, cam, depth, l) for l in range(4)]
        cur, prev = pyr(frames[1][0]), pyr(frames[0][0])
        vcs, ncs = [ctx.upload(v) for v, _ in cur], [ctx.upload(n) for _, n in cur]
        g = [oracle_mod.transform_maps(v, n, np.eye(3), t0) for v, n in prev]
        vgs, ngs = [ctx.upload(v) for v, _ in g], [ctx.upload(n) for _, n in g]
        its = [4, 3, 2, 0]
        track = lambda: ctx.icp_track(vcs, ncs, vgs, ngs, cols, rows, gi, R, t0, its, dist, ang)
        good = track()
        _handoff_fault(ctx, sum(its) - 1, 1, want_dirty=False)
        with pytest.raises(abi.KtError, match="timed out"):
            track()
        assert _handoff_fault(ctx, 0, 0) == 0
        again = track()
        for x, y in zip(good, again):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    finally:
        _handoff_fault(ctx, 0, 0, spin_limit=1 << 22, want_dirty=False)


@pytest.mark.parametrize("levels", [1, 0])
def test_tracker_recovers_from_a_handoff_timeout(ctx, small_scene, levels):
    """The tracker's form of the same (round 6: a time-out is no longer the end of the frame).  The reference's icpStep is stream-ordered and cannot
    time out (reduce.cu:347-419); here the last odometry launch of a frame loses a publisher -> its sweep gives up, the level kernel aborts the
    launch (the waiting workgroups leave), the set-up kernel parks the fusion, and complete_frame re-runs the frame's odometry one launch per
    iteration from the frame's starting pose: NO error, the frame's pose and every later pose and both volumes equal to an undisturbed run's,
    the fallback counted, the level form off for the frames behind it.  A fault that also hits the re-run IS reported; the buffer is clean
    afterwards and a reset sequence gives the undisturbed results again.  Both forms of the ICP chain."""
    from kintinuous_amd import abi
    cam, frames, traj = small_scene
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)

    def make():
        abi._chk(abi.lib().kt_debug_icp_levels(levels))
        try:
            return abi.Tracker(ctx, cfg)
        finally:
            abi._chk(abi.lib().kt_debug_icp_levels(-1))

    def run(trk, fault_at=None, fault=None):
        poses = []
        for k, (d, rgb) in enumerate(frames[:5]):
            if k == fault_at:
                trk.pose()
                fault()
            trk.process_frame_host(d, rgb, 33333 * k)
            poses.append(np.concatenate([x.ravel() for x in trk.pose()]))
        return np.array(poses), trk.volume().copy(), trk.color_volume().copy()

    form = lambda t: abi.lib().kt_tracker_debug_icp_levels(t.h)
    trk = make()
    try:
        P0, V0, C0 = run(trk)
        assert trk.odometry_fallbacks() == 0 and form(trk) == levels
        trk.reset()
        # frame 2: its launch (ONE for the whole frame in the level form; the last of 10 + 5 + 4 otherwise) loses workgroup 0
        P1, V1, C1 = run(trk, 2, lambda: _handoff_fault(ctx, 0 if levels else 18, 1, spin_limit=64, want_dirty=False))
        assert trk.odometry_fallbacks() == 1
        assert np.array_equal(P0.view(np.uint32), P1.view(np.uint32)) and np.array_equal(V0, V1) and np.array_equal(C0, C1)
        assert form(trk) == 0          # demoted: the frames behind a fallback run the stepwise chain
        assert _handoff_fault(ctx, 0, 0) == 0
        trk.close()
        # every launch from frame 2 on loses the publisher: the re-run gives up as well -> reported
        trk = make()
        for k in range(2):
            trk.process_frame_host(frames[k][0], frames[k][1], 33333 * k)
        trk.pose()
        _handoff_fault(ctx, 0, 1000, spin_limit=64, want_dirty=False)
        trk.process_frame_host(frames[2][0], frames[2][1], 33333 * 2)
        with pytest.raises(abi.KtError, match="timed out"):
            trk.pose()
        assert _handoff_fault(ctx, 0, 0) == 0
        trk.reset()
        P2, V2, C2 = run(trk)
        assert np.array_equal(P0.view(np.uint32), P2.view(np.uint32)) and np.array_equal(V0, V2) and np.array_equal(C0, C2)
    finally:
        _handoff_fault(ctx, 0, 0, spin_limit=1 << 22, want_dirty=False)
        trk.close()


def test_hand_off_waits_are_bounded_in_time(ctx, small_scene):
    """The bound that decides in practice is wall time (s_memrealtime), not a number of looks: with the look limit left at its default
    (2^22 agent-scope round trips: seconds) a launch that loses a publisher must give up within the time limit -- lowered here to 2 ms so that
    the test can tell the two apart -- and the frame must still complete through the stepwise re-run."""
    import time
    from kintinuous_amd import abi
    cam, frames, traj = small_scene
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
    abi._chk(abi.lib().kt_debug_icp_levels(1))   # the level form (this small view is "dense" by the tracker's rule and would run stepwise)
    try:
        trk = abi.Tracker(ctx, cfg)
    finally:
        abi._chk(abi.lib().kt_debug_icp_levels(-1))
    try:
        for k in range(2):
            trk.process_frame_host(frames[k][0], frames[k][1], 33333 * k)
        trk.pose()
        abi._chk(abi.lib().kt_debug_wait_limit(ctx.h, 200000))   # 2 ms
        _handoff_fault(ctx, 0, 1, want_dirty=False)             # look limit untouched
        t0 = time.perf_counter()
        trk.process_frame_host(frames[2][0], frames[2][1], 33333 * 2)
        trk.pose()
        dt = time.perf_counter() - t0
        assert trk.odometry_fallbacks() == 1
        assert dt < 0.5, dt    # 2 ms of waiting + the re-run; the look limit alone would be seconds
    finally:
        abi._chk(abi.lib().kt_debug_wait_limit(ctx.h, 0))
        _handoff_fault(ctx, 0, 0, want_dirty=False)
        trk.close()
