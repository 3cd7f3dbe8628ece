"""GPU parity: whole-frame pipeline (KintinuousTracker::processFrame, SURVEY row a16 + a7/a10) through the C-ABI tracker
vs the oracle tracker on the same synthetic frames.
Every kernel is bit-identical to the oracle and the reductions use the reference's summation order, so the only
possible divergence is the device libm (double sin / cos in Rodrigues) -- practically none.  Bars: per-frame pose within
1e-6, identical shift decisions, TSDF / colour volumes byte-identical, slices equal in size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfgs(cam, N, **kw):
    from kintinuous_amd import abi
    from oracle import oracle
    d = dict(volume_size=6.0, voxel_shift=14, overlap=2, static_mode=0, use_rgbd=0, use_rgbd_icp=0, fast_odometry=0, disable_color_angle=0,
             dynamic_cube=0, place_recognition=0)
    d.update(kw)
    g = abi.TrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, d["volume_size"], d["voxel_shift"], d["overlap"], d["static_mode"],
                          d["use_rgbd"], d["use_rgbd_icp"], d["fast_odometry"], d["disable_color_angle"], 0, d["dynamic_cube"], d["place_recognition"])
    o = oracle.OTrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, d["volume_size"], d["voxel_shift"], d["overlap"], d["static_mode"],
                              d["use_rgbd"], d["use_rgbd_icp"], d["fast_odometry"], d["disable_color_angle"], 0, d["dynamic_cube"],
                              d["place_recognition"])
    return g, o


def _run_pair(ctx, cam, frames, N, **kw):
    from kintinuous_amd import abi
    from oracle import oracle
    g, o = _cfgs(cam, N, **kw)
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    max_t, max_r = 0.0, 0.0
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame_host(d, rgb, 33333 * k)
        otr.process_frame(d, rgb, 33333 * k)
        R, t, gc = trk.pose()
        Ro, to, go = otr.pose()
        max_t = max(max_t, float(np.abs(t - to).max() / max(1.0, np.abs(to).max())), float(np.abs(gc - go).max()))
        max_r = max(max_r, float(np.abs(R - Ro).max()))
        assert np.array_equal(trk.voxel_wrap(), otr.voxel_wrap()), f"frame {k}: shift decisions diverged"
    return trk, otr, max_t, max_r


def _volume_close(trk, otr):
    """TSDF and colour volumes (r, g, b and the weight in .w) byte-identical."""
    v, ov = trk.volume(), otr.volume()
    c, oc = trk.color_volume(), otr.color_volume()
    touched = int((oc[..., 3] != 0).sum())
    assert touched > 0
    assert np.array_equal(v, ov), f"tsdf mismatches {int((v != ov).sum())}/{touched}"
    for ch in range(4):
        assert np.array_equal(c[..., ch], oc[..., ch]), f"colour channel {ch} mismatches {int((c[..., ch] != oc[..., ch]).sum())}/{touched}"
    return 0, touched


def _same_points(p, q):
    """Two extracted clouds hold the same points with the same colours (the append order of the extraction is free)."""
    if len(p) != len(q):
        return False
    a = np.sort(np.ascontiguousarray(p).view(np.dtype((np.void, p.dtype.itemsize))).ravel())
    b = np.sort(np.ascontiguousarray(q).view(np.dtype((np.void, q.dtype.itemsize))).ravel())
    return bool(np.array_equal(a, b))


def test_icp_orbit(ctx, oracle_mod, small_scene):
    cam, frames, traj = small_scene
    trk, otr, max_t, max_r = _run_pair(ctx, cam, frames, 96)
    assert max_t < 1e-6 and max_r < 1e-6, (max_t, max_r)
    assert trk.num_poses() == otr.num_poses() == len(frames)
    ts, P, loop = trk.dense_pose(0)
    ts2, P2, loop2 = otr.dense_pose(0)
    assert (ts, loop) == (ts2, loop2) and np.array_equal(P, P2)
    _volume_close(trk, otr)
    # predicted maps of the last frame (raycast with the fused resize pyramid): valid pixels bit-identical
    for lvl in range(4):
        for a, b in ((trk.vmap_g_prev(lvl), otr.vmap_g_prev(lvl)), (trk.nmap_g_prev(lvl), otr.nmap_g_prev(lvl))):
            rows = a.shape[0] // 3
            va, vb = np.isfinite(a[:rows]), np.isfinite(b[:rows])
            assert np.array_equal(va, vb), f"level {lvl}: validity masks differ"
            assert va.sum() > 0
            for p in range(3):
                assert np.array_equal(a[p * rows:(p + 1) * rows][va].view(np.uint32), b[p * rows:(p + 1) * rows][va].view(np.uint32)), (lvl, p)
    # tracking itself must be right: against ground truth (volume frame = scene + 3 m)
    R, t, _ = trk.pose()
    Rg, cg = traj[len(frames) - 1]
    assert np.abs(t - (cg + 3.0)).max() < 0.02 and np.abs(R - Rg).max() < 0.01
    trk.close(); otr.close()


@pytest.mark.parametrize("mode", ["rgbd_icp", "rgbd", "fast"])
def test_rgbd_modes(ctx, oracle_mod, small_scene, mode):
    cam, frames, traj = small_scene
    kw = dict(use_rgbd_icp=1) if mode == "rgbd_icp" else dict(use_rgbd=1) if mode == "rgbd" else dict(fast_odometry=1)
    trk, otr, max_t, max_r = _run_pair(ctx, cam, frames[:5], 64, **kw)
    assert max_t < 1e-6 and max_r < 1e-6, (max_t, max_r)
    _volume_close(trk, otr)
    trk.close(); otr.close()


def test_shifting_crabwalk(ctx, oracle_mod):
    """Wall scan with a coarse shift threshold so that X+ and X- shifts (slab extract + clear) happen within a few frames."""
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    idx = list(range(0, 40, 2)) + list(range(40, 0, -2))  # 30 mm steps out and back
    frames = [synth.render(scene, cam, *traj[i]) for i in idx]
    trk, otr, max_t, max_r = _run_pair(ctx, cam, frames, 96, volume_size=7.0, voxel_shift=3)
    assert max_t < 1e-6 and max_r < 1e-6, (max_t, max_r)
    assert trk.num_slices() == otr.num_slices() and trk.num_slices() >= 4
    dims = set()
    for i in range(trk.num_slices()):
        p, dim = trk.slice(i)
        q, dim2 = otr.slice(i)
        dims.add(dim)
        assert dim == dim2
        assert _same_points(p, q), (i, len(p), len(q))
    assert {0, 1} <= dims  # XPlus and XMinus
    _volume_close(trk, otr)
    trk.finalise(); otr.finalise()
    p, dim = trk.slice(trk.num_slices() - 1)
    q, _ = otr.slice(otr.num_slices() - 1)
    assert dim == 7 and _same_points(p, q)
    trk.close(); otr.close()


def test_static_mode_and_reset(ctx, oracle_mod):
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("farwall")
    traj = synth.static_trajectory(4)
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    trk, otr, max_t, max_r = _run_pair(ctx, cam, frames, 64, static_mode=1)
    assert max_t < 1e-6 and max_r < 1e-6
    assert trk.num_slices() == 0
    _volume_close(trk, otr)
    trk.reset()
    assert trk.num_poses() == 0 and not trk.volume().any()
    trk.close(); otr.close()


def test_device_frames_and_counts(ctx, oracle_mod, small_scene):
    """Device-resident inputs (the bench path) + the U / S counters against the oracle's counts."""
    from kintinuous_amd import abi
    from oracle import oracle
    cam, frames, _ = small_scene
    g, o = _cfgs(cam, 64)
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    trk.enable_counts(True)
    trk.enable_profiling(2)
    for k, (d, rgb) in enumerate(frames[:3]):
        trk.process_frame(ctx.upload(d), ctx.upload(rgb), k)
        otr.process_frame(d, rgb, k)
        U, S = trk.last_counts()
        Uo, So = otr.last_counts()
        assert abs(U - Uo) <= max(4, 1e-3 * Uo)
        if k > 0:
            assert abs(S - So) <= max(16, 1e-3 * So)
    ms = trk.stage_ms()
    assert ms["integrate"][1] >= 2 and ms["integrate"][0] > 0 and ms["tsdf23"][1] >= 0
    trk.close(); otr.close()


@pytest.mark.parametrize("rgbd_icp", [0, 1])
def test_readahead_is_transparent(ctx, small_scene, rgbd_icp):
    """kt_tracker_prefetch_frame (pose-independent stages of the next frame on a second stream) must not change anything:
    same poses, same volumes, same predicted maps -- with in-order read-ahead, with an abandoned read-ahead, and mixed
    with host-frame calls."""
    from kintinuous_amd import abi
    cam, frames, _ = small_scene
    frames = [(np.ascontiguousarray(d, np.uint16), np.ascontiguousarray(rgb, np.uint8)) for d, rgb in frames[:8]]
    dev = [(ctx.upload(d), ctx.upload(rgb)) for d, rgb in frames]
    g, _ = _cfgs(cam, 96, use_rgbd_icp=rgbd_icp)   # the RGB-D inputs ride in the frame sets too

    def run(mode):
        trk = abi.Tracker(ctx, g)
        for k in range(len(dev)):
            if mode == "ahead" and k + 1 < len(dev) and k > 0:
                trk.prefetch_frame(*dev[k + 1])            # before frame k is handed over (frame 1 is built inline)
            if mode == "chaos":
                if k == 2:
                    trk.prefetch_frame(*dev[5])            # announced three frames early: frames 2..4 are built inline meanwhile
                elif k == 4:
                    trk.prefetch_frame(*dev[6])            # two outstanding, consumed in order later
                    with pytest.raises(abi.KtError):
                        trk.prefetch_frame(*dev[7])        # a third is refused
                elif k == 7:
                    trk.prefetch_frame(*dev[1])            # never consumed
            if mode == "host_ahead":
                if k + 1 < len(frames):
                    trk.prefetch_frame_host(frames[k + 1][0], frames[k + 1][1])   # upload + pose-independent stages on the read-ahead stream
                trk.process_frame_host(frames[k][0], frames[k][1], 33333 * k)
            elif mode == "host":
                trk.process_frame_host(frames[k][0], frames[k][1], 33333 * k)
            elif mode == "chaos" and k == 3:
                trk.process_frame_host(frames[k][0], frames[k][1], 33333 * k)
            else:
                trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
            if mode == "late" and k + 1 < len(dev):
                trk.prefetch_frame(*dev[k + 1])            # after frame k has been handed over
        poses = [trk.dense_pose(i)[1].copy() for i in range(trk.num_poses())]
        out = (poses, trk.volume().copy(), trk.color_volume().copy(), [trk.vmap_g_prev(l).copy() for l in range(4)])
        trk.close()
        return out

    ref = run("plain")
    for mode in ("ahead", "late", "chaos", "host", "host_ahead"):
        got = run(mode)
        assert len(got[0]) == len(ref[0])
        for a, b in zip(got[0], ref[0]):
            assert np.array_equal(a, b), mode
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), mode
        for a, b in zip(got[3], ref[3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), mode


@pytest.mark.parametrize("readahead", [0, 1])
def test_ground_truth_odometry(ctx, oracle_mod, readahead):
    """-p mode (GroundTruthOdometry.cpp, KintinuousTracker::loadTrajectory): poses come from a trajectory instead of ICP.  The
    pose arithmetic is host float code on both sides, so the poses are bit-identical; volumes, shifts and slices as in the ICP
    tests.  One frame has no trajectory entry (dropped by preRun), the walk shifts the volume, and with readahead every frame --
    the dropped one too -- is announced one call early."""
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    idx = list(range(0, 40, 2)) + list(range(40, 0, -2))
    poses = [traj[i] for i in idx]
    frames = [tuple(np.ascontiguousarray(a) for a in synth.render(scene, cam, *p)) for p in poses]
    stamps = np.array([33333 * (k + 1) for k in range(len(frames))], np.uint64)
    rows = synth.ground_truth_rows(poses)
    keep = [k for k in range(len(frames)) if k != 7]
    g, o = _cfgs(cam, 96, volume_size=7.0, voxel_shift=3)
    trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
    trk.load_trajectory(stamps[keep], rows[keep])
    otr.load_trajectory(stamps[keep], rows[keep])
    for k, (d, rgb) in enumerate(frames):
        if readahead and k + 1 < len(frames):
            trk.prefetch_frame_host(*frames[k + 1])
        trk.process_frame_host(d, rgb, int(stamps[k]))
        otr.process_frame(d, rgb, int(stamps[k]))
        assert trk.num_poses() == otr.num_poses() == (k + 1 if k < 7 else k)
        R, t, gc = trk.pose()
        Ro, to, go = otr.pose()
        assert np.array_equal(R, Ro) and np.array_equal(t, to) and np.array_equal(gc, go), k
        assert np.array_equal(trk.voxel_wrap(), otr.voxel_wrap()), k
    for i in range(trk.num_poses()):
        a, b = trk.dense_pose(i), otr.dense_pose(i)
        assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1], b[1])
    # the poses follow the walk (volume frame = scene + volume_size / 2)
    _, t, _ = trk.pose()
    wrap = np.asarray(trk.voxel_wrap(), np.float64) * (7.0 / 96)
    assert np.abs(t + wrap - (poses[-1][1] - poses[0][1] + 3.5)).max() < 1e-4
    assert trk.num_slices() == otr.num_slices() and trk.num_slices() >= 4
    for i in range(trk.num_slices()):
        p, dim = trk.slice(i)
        q, dim2 = otr.slice(i)
        assert dim == dim2 and _same_points(p, q)
    _volume_close(trk, otr)
    a, b = trk.vmap_g_prev(0), otr.vmap_g_prev(0)
    va = np.isfinite(a)
    assert np.array_equal(va, np.isfinite(b)) and va.sum() > 0 and np.array_equal(a[va].view(np.uint32), b[va].view(np.uint32))
    trk.close(); otr.close()


@pytest.mark.parametrize("mode", ["icp", "rgbd_icp", "static"])
def test_dynamic_cube(ctx, oracle_mod, mode):
    """-d: the camera turns on the spot; the cube's corner follows the heading (repositionCube), which trips volume shifts.  The pose
    is observed on the host before the frame is fused, so nothing is speculated -- results must still equal the oracle's exactly."""
    from kintinuous_amd import synth
    from scipy.spatial.transform import Rotation
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("room")
    yaws = [0.015 * k for k in range(14)] + [0.015 * (13 - k) for k in range(1, 8)]
    frames = [synth.render(scene, cam, Rotation.from_euler("y", a).as_matrix(), np.zeros(3)) for a in yaws]
    kw = dict(use_rgbd_icp=1) if mode == "rgbd_icp" else dict(static_mode=1) if mode == "static" else {}
    trk, otr, max_t, max_r = _run_pair(ctx, cam, frames, 96, volume_size=6.0, voxel_shift=3, dynamic_cube=1, **kw)
    assert max_t == 0.0 and max_r == 0.0
    assert np.array_equal(trk.volume_basis(), otr.volume_basis())
    if mode == "static":   # parked: the reposition threshold is 3 N voxels, the cube never moves (KintinuousTracker.cpp:403)
        assert np.allclose(trk.volume_basis(), [3.0, 3.0, -0.45], atol=1e-6) and trk.num_slices() == 0
    else:
        assert not np.array_equal(trk.volume_basis(), np.array([3.0, 3.0, 0.0], np.float32))
        assert trk.num_slices() == otr.num_slices() and trk.num_slices() >= 2
        for i in range(trk.num_slices()):
            p, dim = trk.slice(i)
            q, dim2 = otr.slice(i)
            assert dim == dim2 and _same_points(p, q)
    _volume_close(trk, otr)
    trk.close(); otr.close()


def test_errors_are_reported_not_swallowed(ctx, small_scene):
    """The C-ABI fails loudly: bad configurations and bad calls return a status with a message (the ctypes binding raises)."""
    from kintinuous_amd import abi
    cam, frames, _ = small_scene
    d = dict(volume_size=6.0, voxel_shift=14, overlap=2, static_mode=0, use_rgbd=0, use_rgbd_icp=0, fast_odometry=0, disable_color_angle=0)
    mk = lambda cols, rows, N, size: abi.TrackerConfig(cols, rows, N, cam.fx, cam.fy, cam.cx, cam.cy, size, 14, 2, 0, 0, 0, 0, 0, 0)
    for bad in (mk(100, 120, 64, 6.0), mk(160, 120, 0, 6.0), mk(160, 120, 64, -1.0), mk(160, 120, 4096, 6.0)):
        with pytest.raises(abi.KtError) as e:
            abi.Tracker(ctx, bad)
        assert "bad argument" in str(e.value)
    trk = abi.Tracker(ctx, mk(cam.cols, cam.rows, 64, 6.0))
    with pytest.raises(abi.KtError):
        trk.process_frame(0, 0, 0)                      # null device pointers
    with pytest.raises(abi.KtError):
        trk.dense_pose(5)                               # no such pose yet
    trk.process_frame_host(frames[0][0], frames[0][1], 0)
    assert trk.num_poses() == 1
    trk.close()


def test_reset_replays_identically(ctx, small_scene):
    """KintinuousTracker::reset (KintinuousTracker.cpp:262-354): after a reset the same frames give the same poses, volumes and colour
    weights as on a fresh tracker (the colour-weight carry and the wrap copy start over too)."""
    from kintinuous_amd import abi
    cam, frames, _ = small_scene
    g, _ = _cfgs(cam, 64)
    trk = abi.Tracker(ctx, g)

    def run():
        for k, (d, rgb) in enumerate(frames[:6]):
            trk.process_frame_host(d, rgb, 33333 * k)
        return (np.stack([trk.dense_pose(i)[1] for i in range(trk.num_poses())]), trk.volume().copy(), trk.color_volume().copy())

    a = run()
    trk.reset()
    assert trk.num_poses() == 0 and trk.num_slices() == 0
    b = run()
    assert len(a[0]) == len(b[0]) == 6
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    trk.close()


@pytest.mark.parametrize("mode", ["icp", "rgbd_icp"])
def test_place_recognition_tap(ctx, oracle_mod, mode):
    """the loop-closure input tap (KintinuousTracker.cpp:601-624, 706-717, 917-958, 1035-1045) on a shifting
    crab-walk: which frames are sampled (by movement, or taken along by a volume shift), their stored pose, the isLoopPose flags of
    the dense pose graph and the sample each slice carries -- identical to the oracle."""
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    frames = [synth.render(scene, cam, *traj[i]) for i in range(0, 90)]
    trk, otr, max_t, max_r = _run_pair(ctx, cam, frames, 96, volume_size=7.0, voxel_shift=3, use_rgbd_icp=int(mode == "rgbd_icp"), place_recognition=1)
    trk.finalise(); otr.finalise()
    assert max_t < 1e-5 and max_r < 1e-5
    assert trk.num_poses() == otr.num_poses()
    gl = [trk.dense_pose(i)[2] for i in range(trk.num_poses())]
    ol = [otr.dense_pose(i)[2] for i in range(otr.num_poses())]
    assert gl == ol and sum(ol) >= 5 and not all(ol)
    gs, os_ = trk.pr_samples(), otr.pr_samples()
    assert len(gs) == len(os_) == sum(ol) + 1            # + the final slice's sample
    for a, b in zip(gs, os_):
        assert a[0] == b[0] and np.abs(a[1] - b[1]).max() < 1e-5 and np.abs(a[2] - b[2]).max() < 1e-5
    assert [gs[i][3] for i in range(len(gs) - 1)] == [i for i, f in enumerate(ol) if f]     # sample i belongs to the i-th loop pose
    assert trk.num_slices() == otr.num_slices() >= 5
    ids = [trk.slice_pr_id(i) for i in range(trk.num_slices())]
    assert ids == [otr.slice_pr_id(i) for i in range(otr.num_slices())]
    assert ids[-1] == len(gs) - 1 and any(i >= 0 for i in ids[:-1])


@pytest.mark.parametrize("shared_ctx", [True, False])
def test_two_trackers_interleaved(ctx, small_scene, shared_ctx):
    """Two trackers in one process, fed alternately (one ICP, one RGB-D + ICP with read-ahead) -- on one context (one stream: the scratch
    buffers, reduction granules and epochs are shared) and on two contexts (two streams running concurrently on the GPU).  Each must
    produce exactly what it produces alone: nothing of a tracker's state may live in the context or in the library's globals."""
    from kintinuous_amd import abi
    cam, frames, traj = small_scene

    def solo(c, kw, seq):
        g, _ = _cfgs(cam, 64, **kw)
        t = abi.Tracker(c, g)
        for k, (d, rgb) in enumerate(seq):
            t.process_frame_host(d, rgb, 33333 * k)
        out = ([t.dense_pose(i)[1].copy() for i in range(t.num_poses())], t.volume().copy(), t.color_volume().copy())
        t.close()
        return out

    seq_a, seq_b = frames[:6], frames[1:7][::-1]
    want_a, want_b = solo(ctx, {}, seq_a), solo(ctx, dict(use_rgbd_icp=1), seq_b)
    ctx2 = ctx if shared_ctx else abi.Ctx(0)
    ga, _ = _cfgs(cam, 64)
    gb, _ = _cfgs(cam, 64, use_rgbd_icp=1)
    ta, tb = abi.Tracker(ctx, ga), abi.Tracker(ctx2, gb)
    for k in range(6):
        if k + 1 < 6:
            tb.prefetch_frame_host(*seq_b[k + 1])
        ta.process_frame_host(seq_a[k][0], seq_a[k][1], 33333 * k)
        tb.process_frame_host(seq_b[k][0], seq_b[k][1], 33333 * k)
    for t, want in ((ta, want_a), (tb, want_b)):
        poses = [t.dense_pose(i)[1] for i in range(t.num_poses())]
        assert len(poses) == len(want[0])
        for p, q in zip(poses, want[0]):
            assert np.array_equal(p, q)
        assert np.array_equal(t.volume(), want[1]) and np.array_equal(t.color_volume(), want[2])
    ta.close(); tb.close()
    if not shared_ctx:
        ctx2.close()


def test_slice_downloads_survive_reset_and_destroy(ctx):
    """Slices are downloaded by helper threads behind an event (fetch_slice).  Shifts in quick succession, a reset and a destroy while
    downloads may still be in flight, slices read out of order: nothing may hang, crash or hand out a half-filled slice."""
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    idx = list(range(0, 40, 2)) + list(range(40, 0, -2))   # 30 mm steps out and back (test_shifting_crabwalk): X+ and X- slabs with points
    frames = [synth.render(scene, cam, *traj[i]) for i in idx]
    g, _ = _cfgs(cam, 96, volume_size=7.0, voxel_shift=3)

    def run(read_between):
        t = abi.Tracker(ctx, g)
        sizes = []
        for k, (d, rgb) in enumerate(frames):
            t.process_frame_host(d, rgb, 33333 * k)
            if read_between and k % 7 == 0 and t.num_slices():
                sizes.append(len(t.slice(t.num_slices() - 1)[0]))
        t.finalise()                            # the whole volume as one more, large slice
        n = t.num_slices()
        pts = [t.slice(i)[0] for i in reversed(range(n))]
        return t, n, pts[::-1]

    t, n, pts = run(False)
    assert n >= 4
    # the oracle says what each slice holds (slabs at the cube's unobserved side faces are legitimately empty)
    from oracle import oracle
    _, o = _cfgs(cam, 96, volume_size=7.0, voxel_shift=3)
    otr = oracle.OracleTracker(o)
    for k, (d, rgb) in enumerate(frames):
        otr.process_frame(d, rgb, 33333 * k)
    otr.finalise()
    assert otr.num_slices() == n
    for i in range(n):
        assert _same_points(pts[i], otr.slice(i)[0]), i
    assert len(pts[-1]) > 1000
    otr.close()
    t.reset()                                   # joins whatever is still in flight, drops the slices
    assert t.num_slices() == 0
    for k, (d, rgb) in enumerate(frames[:12]):  # shifts again ...
        t.process_frame_host(d, rgb, 33333 * k)
    t.close()                                   # ... and is destroyed with the last downloads possibly unfinished
    t2, n2, pts2 = run(True)
    assert n2 == n and all(_same_points(a, b) for a, b in zip(pts, pts2))
    t2.close()


@pytest.mark.parametrize("mode", ["icp", "rgbd_icp"])
def test_planned_voxel_pass_is_transparent(ctx, oracle_mod, monkeypatch, mode):
    """Planning ahead (csrc/kt_volume.hip): with read-ahead the voxel kernel's task plan is made for a PREDICTED pose while the odometry
    iterates; the set-up kernel accepts it when the pose lands inside the plan's margins and parks the frame for the in-stream pre-pass
    otherwise.  A shifting sequence through (a) plans that hit, (b) plans forced to miss (margins scaled to 0) and (c) no planning must
    give the same poses, volumes, colour volumes and slices -- and all three the oracle's."""
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    idx = list(range(0, 40, 2)) + list(range(40, 20, -2))
    frames = [synth.render(scene, cam, *traj[i]) for i in idx]
    frames = [(np.ascontiguousarray(d, np.uint16), np.ascontiguousarray(rgb, np.uint8)) for d, rgb in frames]
    dev = [(ctx.upload(d), ctx.upload(rgb)) for d, rgb in frames]
    g, o = _cfgs(cam, 96, volume_size=7.0, voxel_shift=3, use_rgbd_icp=int(mode == "rgbd_icp"))

    def run(env):
        for k in ("KT_NO_PLAN", "KT_PLAN_MARGIN_SCALE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        trk = abi.Tracker(ctx, g)
        for k in range(len(dev)):
            if k + 1 < len(dev):
                trk.prefetch_frame(*dev[k + 1])
            trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
        out = dict(poses=[trk.dense_pose(i)[1].copy() for i in range(trk.num_poses())], vol=trk.volume().copy(), col=trk.color_volume().copy(),
                   slices=[trk.slice(i) for i in range(trk.num_slices())], stats=trk.plan_stats(), wrap=[int(v) for v in trk.voxel_wrap()])
        trk.close()
        return out

    hit, miss, off = run({}), run({"KT_PLAN_MARGIN_SCALE": "0"}), run({"KT_NO_PLAN": "1"})
    # (the first frames have no pose history, a volume shift drops the plan made before it, and the turn-around of this walk -- 30 mm
    # steps out, then back -- is a prediction the margins reject: those frames take the in-stream pre-pass)
    assert hit["stats"][0] >= 10 and hit["stats"][0] > 2 * hit["stats"][1], hit["stats"]
    assert miss["stats"][0] == 0 and miss["stats"][1] >= 10, miss["stats"]
    assert off["stats"] == (0, 0)
    assert len(hit["slices"]) >= 2
    for other in (miss, off):
        assert len(other["poses"]) == len(hit["poses"]) and all(np.array_equal(a, b) for a, b in zip(other["poses"], hit["poses"]))
        assert np.array_equal(other["vol"], hit["vol"]) and np.array_equal(other["col"], hit["col"]) and other["wrap"] == hit["wrap"]
        assert len(other["slices"]) == len(hit["slices"])
        for (p, d1), (q, d2) in zip(other["slices"], hit["slices"]):
            assert d1 == d2 and _same_points(p, q)
    otr = oracle.OracleTracker(o)
    for k, (d, rgb) in enumerate(frames):
        otr.process_frame(d, rgb, 33333 * k)
    assert np.array_equal(hit["vol"], otr.volume()) and np.array_equal(hit["col"], otr.color_volume())
    otr.close()


def test_plan_is_bound_to_its_frame(ctx, monkeypatch):
    """A task plan is made from ONE read-ahead frame's depth (its pixel records and tile maxima) and may only serve that frame.  The
    caller is allowed to skip a read-ahead: announce A and B, process B.  The plan made for "the next frame" (A) must not be used for
    B -- its column intervals would end where A's surfaces end -- and the skipped set must not be rewritten under the plan's kernels.
    B is A's view with every surface 0.4 m farther away, so a plan from A's depth cuts B's updates short; the result must equal a run
    without planning."""
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(12)
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    frames = [(np.ascontiguousarray(d, np.uint16), np.ascontiguousarray(rgb, np.uint8)) for d, rgb in frames]
    far = [(np.where(d > 0, d + 400, 0).astype(np.uint16), rgb) for d, rgb in frames]
    dev = [(ctx.upload(d), ctx.upload(rgb)) for d, rgb in frames]
    dev_far = [(ctx.upload(d), ctx.upload(rgb)) for d, rgb in far]
    g, _ = _cfgs(cam, 96)

    def run(env):
        monkeypatch.delenv("KT_NO_PLAN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        trk = abi.Tracker(ctx, g)
        for k in range(6):   # history for the motion model, plans that hit
            trk.prefetch_frame(*dev[k + 1])
            trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
        # frame 6 (A) is read ahead and has been planned for; now announce a SECOND frame (B) and process that one
        trk.prefetch_frame(*dev_far[7])
        trk.process_frame(dev_far[7][0], dev_far[7][1], 33333 * 6)     # ... skipping A
        trk.prefetch_frame(*dev[8])
        trk.process_frame(dev[7][0], dev[7][1], 33333 * 7)             # an unannounced frame while frame 8 is pending (and planned for)
        trk.prefetch_frame(*dev[9])
        trk.process_frame(dev[8][0], dev[8][1], 33333 * 8)             # back to the regular order
        trk.process_frame(dev[9][0], dev[9][1], 33333 * 9)
        out = dict(vol=trk.volume().copy(), col=trk.color_volume().copy(), poses=[trk.dense_pose(i)[1].copy() for i in range(trk.num_poses())],
                   stats=trk.plan_stats())
        trk.close()
        return out

    plan, off = run({}), run({"KT_NO_PLAN": "1"})
    assert plan["stats"][0] >= 1 and off["stats"] == (0, 0), plan["stats"]
    assert len(plan["poses"]) == len(off["poses"]) and all(np.array_equal(a, b) for a, b in zip(plan["poses"], off["poses"]))
    assert np.array_equal(plan["vol"], off["vol"]) and np.array_equal(plan["col"], off["col"])


def _plan_edge_runs(ctx, cam, frames, N, kw, cases, theta_tau):
    """Run A (no plans) logs the pose every frame arrives at; runs B use those poses, offset by (fr theta, ft tau), as the plans'
    predictions.  Returns {case: (hits, misses, equal)}."""
    from kintinuous_amd import abi
    g, _ = _cfgs(cam, N, **kw)
    dev = [(ctx.upload(np.ascontiguousarray(d, np.uint16)), ctx.upload(np.ascontiguousarray(c, np.uint8))) for d, c in frames]

    def play(trk):
        for k in range(len(dev)):
            if k + 1 < len(dev):
                trk.prefetch_frame(*dev[k + 1])
            trk.process_frame(dev[k][0], dev[k][1], 33333 * k)

    os.environ["KT_NO_PLAN"] = "1"
    try:
        ref = abi.Tracker(ctx, g)
    finally:
        del os.environ["KT_NO_PLAN"]
    ref.pose_log(True)
    play(ref)
    truth = ref.pose_log(fetch=True)
    assert len(truth) == len(frames)
    vol, col, wrap, nsl = ref.volume().copy(), ref.color_volume().copy(), ref.voxel_wrap().copy(), ref.num_slices()
    ref.close()
    out = {}
    for (fr, ft) in cases:
        for (theta, tau) in theta_tau:
            trk = abi.Tracker(ctx, g)
            trk.plan_truth(truth, fr, ft, theta, tau, seed=int(1000 * fr + 10 * ft) + 1)
            play(trk)
            hits, misses = trk.plan_stats()
            equal = bool(np.array_equal(trk.voxel_wrap(), wrap) and trk.num_slices() == nsl and np.array_equal(trk.volume(), vol) and
                         np.array_equal(trk.color_volume(), col))
            trk.close()
            out[(fr, ft, theta, tau)] = (hits, misses, equal)
    return out


import os  # noqa: E402


@pytest.mark.parametrize("size", ["small", "orbit512", "farwall768"])
def test_plan_margins_hold_at_their_edge(ctx, size):
    """The one mechanism whose failure is silent: a plan that is too tight DROPS voxels (skipped iterations do not run at all).  The
    pre-pass is widened for a pose within theta (rotation) and tau (translation) of the prediction; here every frame's pose lands at
    0.9 / 0.98 of BOTH margins in a random direction (the prediction is the frame's real pose from an identical run, offset by exactly
    that much) -- the plans must be accepted and the volumes must equal those of a run without plans, byte for byte; at 1.05 of either
    margin the set-up kernel must reject the plan (and the fall-back gives the same volumes).  Margins: small realistic ones and the
    caps (20 mrad, 20 mm)."""
    from kintinuous_amd import synth
    if size == "small":
        cam = synth.Camera.small(160, 120)
        traj = synth.crabwalk_trajectory(420)
        frames = [synth.render(synth.Scene("wall"), cam, *traj[i]) for i in list(range(0, 40, 2)) + list(range(40, 20, -2))]
        N, kw = 96, dict(volume_size=7.0, voxel_shift=3)
        margins = [(1.0e-3, 3.0e-3), (0.02, 0.02)]
    elif size == "orbit512":
        cam = synth.Camera()
        _, frames, _, kw = synth.sequence("orbit", 24, cam)   # through the orbit's first shift
        N = 512
        margins = [(2.0e-3, 4.0e-3)]
    else:
        cam = synth.Camera.scaled(2)
        _, frames, _, kw = synth.sequence("farwall", 8, cam)
        kw = dict(kw, static_mode=1)
        N = 768
        margins = [(2.0e-3, 4.0e-3)]
    accept = [(0.9, 0.9), (0.9, 0.98), (0.98, 0.9), (0.98, 0.98)]
    reject = [(1.05, 0.5), (0.5, 1.05)]
    res = _plan_edge_runs(ctx, cam, frames, N, kw, accept + reject, margins)
    # planned frames: all but the first three (no motion history yet), minus the frames that shift the volume and the frame behind
    # each shift (its plan was made for the old storage wrap)
    floor = {"small": 8, "orbit512": len(frames) - 3 - 6, "farwall768": len(frames) - 3 - 1}[size]
    for key, (hits, misses, equal) in res.items():
        assert equal, (key, hits, misses)
        if key[:2] in accept:
            assert hits >= floor and misses == 0, (key, hits, misses)     # every plan that was tried was accepted
        else:
            assert hits == 0 and misses >= floor, (key, hits, misses)     # ... and rejected



def test_icp_chain_per_level_equals_per_iteration(ctx, small_scene):
    """The tracker's ICP-only odometry as one launch per pyramid level (kt_icp_level_kernel: the iterations of a level hand the pose over
    inside the kernel) and as one launch per iteration (kt_icp_kernel x 19) are the same arithmetic: every pose and both volumes bit-equal."""
    from kintinuous_amd import abi
    cam, frames, traj = small_scene
    g, _ = _cfgs(cam, 96)
    out = []
    for levels in (0, 1):
        abi._chk(abi.lib().kt_debug_icp_levels(levels))
        try:
            trk = abi.Tracker(ctx, g)
        finally:
            abi._chk(abi.lib().kt_debug_icp_levels(-1))
        poses = []
        for k, (d, rgb) in enumerate(frames):
            trk.process_frame_host(d, rgb, 33333 * k)
            poses.append(np.concatenate([x.ravel() for x in trk.pose()]))
        out.append((np.array(poses), trk.volume().copy(), trk.color_volume().copy()))
        trk.close()
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_ri_chain_per_level_equals_per_iteration(ctx, small_scene):
    """-ri (joint RGB-D + ICP odometry, RGBDOdometry.cpp:165-393): one launch per pyramid level (round 6, kt_joint_level_kernel: the residual pass, the
    grid-wide sigma and both reductions of every iteration inside one resident kernel) and two launches per iteration (kt_residual_kernel +
    kt_joint_kernel) are the same arithmetic: every pose and both volumes bit-equal, with and without read-ahead."""
    from kintinuous_amd import abi
    cam, frames, traj = small_scene
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 1, 0, 0, 0)
    out = []
    for levels, ahead in ((0, False), (1, False), (1, True)):
        abi._chk(abi.lib().kt_debug_icp_levels(levels))
        abi._chk(abi.lib().kt_debug_ri_levels(levels))   # (off by default: measured slower, kt_tracker.hip rgbd_odometry)
        try:
            trk = abi.Tracker(ctx, cfg)
        finally:
            abi._chk(abi.lib().kt_debug_icp_levels(-1))
        dev = [(ctx.upload(d), ctx.upload(rgb)) for d, rgb in frames]
        poses = []
        for k in range(len(frames)):
            if ahead and k + 1 < len(frames):
                trk.prefetch_frame(dev[k + 1][0], dev[k + 1][1])
            trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
            if not ahead:
                poses.append(np.concatenate([x.ravel() for x in trk.pose()]))
        if ahead:
            poses = [trk.dense_pose(i)[1].ravel() for i in range(trk.num_poses())]
        assert abi.lib().kt_tracker_debug_icp_levels(trk.h) == levels and trk.odometry_fallbacks() == 0
        out.append((np.array(poses), trk.volume().copy(), trk.color_volume().copy()))
        trk.close()
        abi._chk(abi.lib().kt_debug_ri_levels(-1))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    for k in (1, 2):
        assert np.array_equal(out[0][1], out[k][1]) and np.array_equal(out[0][2], out[k][2])
    assert np.abs(out[0][0][-1] - out[0][0][0]).max() > 1e-4    # it did track


def test_side_gate_is_transparent(ctx, small_scene):
    """KT_SIDE_GATE (round 6): with the gate on, the read-ahead stream waits for an event behind the voxel kernel of the frame in flight and the main
    stream joins the side streams in front of the next set-up kernel -- a change of WHEN kernels run, never of what they compute.  A sequence with
    read-ahead and volume shifts gives the same poses, slices and volumes with the gate off, on, and on with the odometry one launch per iteration
    (the dense-view default)."""
    import os
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    frames = [synth.render(scene, cam, *traj[i]) for i in range(40)]
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 3, 2, 0, 0, 0, 0, 0, 0)
    dev = [(ctx.upload(d), ctx.upload(rgb)) for d, rgb in frames]
    out = []
    saved = os.environ.get("KT_SIDE_GATE")
    try:
        for gate, levels in (("0", 1), ("1", 1), ("1", 0)):
            os.environ["KT_SIDE_GATE"] = gate
            abi._chk(abi.lib().kt_debug_icp_levels(levels))
            try:
                trk = abi.Tracker(ctx, cfg)
            finally:
                abi._chk(abi.lib().kt_debug_icp_levels(-1))
            assert abi.lib().kt_tracker_debug_side_gate(trk.h) == int(gate)
            for k in range(len(dev)):
                if k + 1 < len(dev):
                    trk.prefetch_frame(dev[k + 1][0], dev[k + 1][1])
                trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
            poses = np.array([trk.dense_pose(i)[1].ravel() for i in range(trk.num_poses())])
            # (the order of the points inside a slice is the extraction kernel's atomic order: compare slices as sorted sets of 32-byte points)
            slices = [b"".join(sorted(bytes(r) for r in trk.slice(i)[0].view(np.uint8).reshape(-1, 32))) for i in range(trk.num_slices())]
            out.append((poses, trk.volume().copy(), trk.color_volume().copy(), slices, trk.voxel_wrap().copy()))
            assert trk.odometry_fallbacks() == 0
            trk.close()
    finally:
        if saved is None:
            os.environ.pop("KT_SIDE_GATE", None)
        else:
            os.environ["KT_SIDE_GATE"] = saved
    assert np.abs(out[0][4]).max() > 0                      # the volume did shift
    for k in (1, 2):
        assert np.array_equal(out[0][0].view(np.uint32), out[k][0].view(np.uint32))
        assert np.array_equal(out[0][1], out[k][1]) and np.array_equal(out[0][2], out[k][2])
        assert out[0][3] == out[k][3] and np.array_equal(out[0][4], out[k][4])


def test_fused_setup_is_transparent(ctx):
    """KT_ICP_FUSED_SETUP=1 (round 6, off by default: measured no faster): the frame's set-up -- final pose, shift decision, plan check, z tables,
    colour-weight carry, the plan's walk checkpoints, the host's mirror -- runs in the epilogue of the odometry's single launch (csrc/kt_setup.hpp)
    instead of in kt_frame_setup_kernel.  Same code, another home: poses, slices and volumes identical on a read-ahead sequence with planned frames
    and volume shifts, and as many plans accepted."""
    import os
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(320, 240)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    frames = [synth.render(scene, cam, *traj[i]) for i in range(40)]
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 256, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 6, 2, 0, 0, 0, 0, 0, 0)   # a sparse view: the level form
    dev = [(ctx.upload(d), ctx.upload(rgb)) for d, rgb in frames]
    out = []
    saved = os.environ.get("KT_ICP_FUSED_SETUP")
    try:
        for fused in ("0", "1"):
            os.environ["KT_ICP_FUSED_SETUP"] = fused
            trk = abi.Tracker(ctx, cfg)
            for k in range(len(dev)):
                if k + 1 < len(dev):
                    trk.prefetch_frame(dev[k + 1][0], dev[k + 1][1])
                trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
            poses = np.array([trk.dense_pose(i)[1].ravel() for i in range(trk.num_poses())])
            slices = [b"".join(sorted(bytes(r) for r in trk.slice(i)[0].view(np.uint8).reshape(-1, 32))) for i in range(trk.num_slices())]
            assert abi.lib().kt_tracker_debug_icp_levels(trk.h) == 1 and trk.odometry_fallbacks() == 0
            out.append((poses, trk.volume().copy(), trk.color_volume().copy(), slices, trk.voxel_wrap().copy(), trk.plan_stats()))
            trk.close()
    finally:
        if saved is None:
            os.environ.pop("KT_ICP_FUSED_SETUP", None)
        else:
            os.environ["KT_ICP_FUSED_SETUP"] = saved
    assert np.abs(out[0][4]).max() > 0 and out[0][5][0] > 10          # the volume shifted, frames were planned
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert out[0][3] == out[1][3] and np.array_equal(out[0][4], out[1][4]) and out[0][5] == out[1][5]


def test_the_level_form_needs_to_be_alone(ctx, small_scene):
    """kt_icp_level_kernel's workgroups wait for each other inside a launch and need the whole machine: a second tracker fed next to it can keep
    its last workgroup out until the bounded waits give up.  So the form is chosen per frame: only while the tracker is the process's only live
    one.  Two trackers fed alternately (their frames overlap on the GPU) both run the launch per iteration and give the same poses, bit for bit,
    as a lone tracker in the level form."""
    import os
    from kintinuous_amd import abi
    if os.environ.get("KT_ICP_LEVELS") == "0":
        pytest.skip("the environment switches the level form off")
    cam, frames, traj = small_scene
    g, _ = _cfgs(cam, 96)
    form = lambda t: abi.lib().kt_tracker_debug_icp_levels(t.h)
    last = lambda t: np.concatenate([x.ravel() for x in t.pose()]).view(np.uint32)

    def make():   # the level form asked for explicitly (round 6: this small view is "dense" by the tracker's rule and would default to the stepwise chain)
        abi._chk(abi.lib().kt_debug_icp_levels(1))
        try:
            return abi.Tracker(ctx, g)
        finally:
            abi._chk(abi.lib().kt_debug_icp_levels(-1))

    probe = make()   # alone, unless an earlier test of this process left a tracker open
    for k in range(2):
        probe.process_frame_host(frames[k][0], frames[k][1], 33333 * k)
    probe.pose()
    alone = form(probe) == 1
    probe.close()
    if not alone:
        pytest.skip("another tracker is alive in this process")
    a = make()
    b = make()
    for k, (d, rgb) in enumerate(frames):
        a.process_frame_host(d, rgb, 33333 * k)
        b.process_frame_host(d, rgb, 33333 * k)
    pa, pb = last(a), last(b)
    assert form(a) == 0 and form(b) == 0 and np.array_equal(pa, pb)
    a.close()
    b.reset()
    for k, (d, rgb) in enumerate(frames):
        b.process_frame_host(d, rgb, 33333 * k)
    assert form(b) == 1 and np.array_equal(last(b), pa)
    b.close()
