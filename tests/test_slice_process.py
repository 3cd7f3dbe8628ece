"""the per-slice stage of the reference's CloudSliceProcessor (backend/CloudSliceProcessor.cpp:87-163) -- weight cull,
pcl::VoxelGrid at the voxel leaf size, pcl::NormalEstimation with the 20 nearest neighbours -- as kt_slice_process on the GPU.
PCL 1.7 is not vendored with the reference and not installed: the oracle restates its published algorithms (PARITY UNPINNED against PCL
itself; the two orders PCL leaves to std::sort and FLANN are fixed, see oracle/kt_oracle_kernels.c).
 CPU : known answers on the oracle -- a tilted plane gives its normal, leaf counts and centroids match a numpy restatement;
 GPU : kt_slice_process against the oracle on real extracted slices: counts, leaf order, positions, colour bytes AND (since round 4)
       normals and curvature exact (until then within 1e-4: cosf / sinf / atan2f of the device library differ from libm in
       the last bits, the single-pass float covariance amplifies that)."""
import numpy as np
import pytest


def _plane_cloud(n=20000, seed=0):
    from oracle.oracle import POINT_DTYPE
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    p = np.zeros(n, POINT_DTYPE)
    p["xyz"][:, 0], p["xyz"][:, 1] = xy[:, 0], xy[:, 1]
    p["xyz"][:, 2] = np.float32(0.2) * xy[:, 0] + np.float32(0.1) * xy[:, 1] + np.float32(3)
    p["bgra"][:] = rng.integers(0, 256, (n, 4))
    return p


def test_oracle_known_answers(oracle_mod):
    p = _plane_cloud()
    cull, leaf = 100, 0.05
    out = oracle_mod.slice_process(p, cull, leaf)
    kept = p[p["bgra"][:, 3] >= cull]
    # VoxelGrid: one output point per occupied leaf, the float mean of its points (in input order), colours truncated, alpha 0
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(kept["xyz"] * inv).astype(np.int64)
    ijk -= np.floor(kept["xyz"].min(axis=0) * inv).astype(np.int64)
    div = ijk.max(axis=0) + 1
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq = np.unique(key)
    assert len(out) == len(uniq)
    order = np.argsort(key, kind="stable")
    first = np.searchsorted(key[order], uniq)
    for q in (0, len(uniq) // 2, len(uniq) - 1):
        idx = order[first[q]: (first[q + 1] if q + 1 < len(uniq) else len(key))]
        acc = np.zeros(3, np.float32)
        col = np.zeros(3, np.float32)
        for i in idx:
            acc += kept["xyz"][i]
            col += kept["bgra"][i, :3].astype(np.float32)
        rn = np.float32(1.0) / np.float32(len(idx))     # Eigen 3.2: `centroid /= n` multiplies by Scalar(1) / n
        assert np.array_equal(out["xyz"][q], acc * rn)
        assert np.array_equal(out["bgra"][q, :3], (col * rn).astype(np.uint8)) and out["bgra"][q, 3] == 0
    assert np.all(out["one"] == 1)
    # NormalEstimation: the plane's normal, pointing at the sensor origin, tiny curvature
    nt = np.array([0.2, 0.1, -1.0]) / np.linalg.norm([0.2, 0.1, -1.0])
    assert np.abs(out["normal"] @ nt - 1).max() < 1e-3
    assert ((out["normal"] * -out["xyz"]).sum(axis=1) > 0).all()
    assert out["curvature"].max() < 1e-3
    # fewer than 3 points: NaN normals; no cull when weightCull is 0
    tiny = oracle_mod.slice_process(p[:2], 0, 0.001)
    assert len(tiny) == 2 and np.isnan(tiny["normal"]).all() and np.isnan(tiny["curvature"]).all()
    assert len(oracle_mod.slice_process(p[:500], 0, 1e-4)) == 500


def _real_slices(ctx):
    """Slices as the tracker extracts them: a crab-walk with shifts (shift slabs) plus the final full-volume cloud."""
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(320, 240)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    frames = [synth.render(scene, cam, *traj[i]) for i in range(60)]
    trk = abi.Tracker(ctx, abi.TrackerConfig(cam.cols, cam.rows, 160, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 6, 2, 0, 0, 0, 0, 0, 0))
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame_host(d, rgb, 33333 * k)
    trk.finalise()
    slices = [trk.slice(i)[0] for i in range(trk.num_slices())]
    trk.close()
    return slices, 7.0 / 160


@pytest.mark.gpu
def test_gpu_matches_oracle_on_extracted_slices(ctx, oracle_mod):
    from kintinuous_amd import abi
    slices, leaf = _real_slices(ctx)
    assert len(slices) >= 3 and max(len(s) for s in slices) > 3000, [len(s) for s in slices]
    checked = 0
    for s in slices:
        if len(s) == 0:
            assert len(abi.slice_process(ctx, s, 0, leaf)) == 0
            continue
        for cull in (0, 2):
            want = oracle_mod.slice_process(s, cull, leaf)
            got = abi.slice_process(ctx, s, cull, leaf)
            assert len(got) == len(want)
            if len(want) == 0:
                continue
            assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["bgra"], want["bgra"]) and np.array_equal(got["one"], want["one"])
            nan = np.isnan(want["normal"]).any(axis=1)
            assert np.array_equal(np.isnan(got["normal"]).any(axis=1), nan)
            ok = ~nan
            # round 4: sin / cos / atan2 of pcl::computeRoots are restated identically on both sides (sp_atan2_pos, sp_sincos): equal bits
            assert np.array_equal(got["normal"][ok], want["normal"][ok]), float(np.abs(got["normal"][ok] - want["normal"][ok]).max())
            assert np.array_equal(got["curvature"][ok], want["curvature"][ok]), float(np.abs(got["curvature"][ok] - want["curvature"][ok]).max())
            checked += int(ok.sum())
    assert checked > 4000


@pytest.mark.gpu
def test_gpu_plane_and_edge_cases(ctx, oracle_mod):
    from kintinuous_amd import abi
    p = _plane_cloud(30000, seed=3)
    for leaf in (0.05, 0.011, 0.5):       # dense (runs of several points per leaf), sparse (the neighbour search widens), 12 leaves in all
        want = oracle_mod.slice_process(p, 100, leaf)
        got = abi.slice_process(ctx, p, 100, leaf)
        assert len(got) == len(want) > 0
        assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["bgra"], want["bgra"])
        ok = ~np.isnan(want["normal"]).any(axis=1)
        assert np.array_equal(~np.isnan(got["normal"]).any(axis=1), ok)
        if ok.any():
            assert np.array_equal(got["normal"][ok], want["normal"][ok]) and np.array_equal(got["curvature"][ok], want["curvature"][ok]), \
                (float(np.abs(got["normal"][ok] - want["normal"][ok]).max()), float(np.abs(got["curvature"][ok] - want["curvature"][ok]).max()))
    got = abi.slice_process(ctx, p[:2], 0, 0.001)
    assert len(got) == 2 and np.isnan(got["normal"]).all()
    got = abi.slice_process(ctx, p[:300], 0, 1e-4)          # "leaf size too small": the cloud passes through unfiltered
    want = oracle_mod.slice_process(p[:300], 0, 1e-4)
    assert len(got) == len(want) == 300 and np.array_equal(got["xyz"], want["xyz"])
    assert len(abi.slice_process(ctx, p[:50], 255, 0.05)) == len(oracle_mod.slice_process(p[:50], 255, 0.05))


@pytest.mark.gpu
def test_stage_behind_the_tracker_shift_path(ctx, oracle_mod):
    """kt_tracker_enable_slice_stage: every extracted slab goes through kt_slice_process_device on the tracker's slice stream -- the
    slab stays on the device, its length is read from the extraction kernel's own counter -- and the processed points travel with the
    slice.  They must be what the host-array entry point makes of the same raw slice (bit for bit: same kernels, same input order),
    hence the oracle's; the raw slices, poses and volumes must not notice the stage."""
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(320, 240)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    frames = [synth.render(scene, cam, *traj[i]) for i in range(50)]
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 160, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 6, 2, 0, 0, 0, 0, 0, 0)

    def run(stage):
        trk = abi.Tracker(ctx, cfg)
        if stage:
            trk.enable_slice_stage(True, weight_cull=2)
        for k, (d, rgb) in enumerate(frames):
            trk.process_frame_host(d, rgb, 33333 * k)
        trk.finalise()
        out = dict(raw=[trk.slice(i) for i in range(trk.num_slices())], proc=[trk.slice_processed(i) for i in range(trk.num_slices())],
                   pose=trk.pose(), vol=trk.volume().copy())
        trk.close()
        return out

    a, b = run(True), run(False)
    assert len(a["raw"]) == len(b["raw"]) >= 3 and all(p is None for p in b["proc"]) and all(p is not None for p in a["proc"])
    assert np.array_equal(a["vol"], b["vol"]) and all(np.array_equal(x, y) for x, y in zip(a["pose"], b["pose"]))
    leaf = 7.0 / 160
    total = 0
    for (raw, dim), proc, (raw_b, dim_b) in zip(a["raw"], a["proc"], b["raw"]):
        assert dim == dim_b and len(raw) == len(raw_b)
        want = abi.slice_process(ctx, raw, 2, leaf)
        assert len(proc) == len(want) and proc.tobytes() == want.tobytes()
        ref = oracle_mod.slice_process(raw.view(oracle_mod.POINT_DTYPE), 2, leaf)
        assert len(ref) == len(proc) and np.array_equal(ref["xyz"], proc["xyz"]) and np.array_equal(ref["bgra"], proc["bgra"])
        total += len(proc)
    assert total > 3000


@pytest.mark.gpu
def test_device_entry_point_counts_on_the_device(ctx, oracle_mod):
    """kt_slice_process_device with a device-resident count that is smaller than the host's bound, and one larger than the buffer (an
    extraction that overflowed its capacity): the stage takes min(count, bound) points."""
    import ctypes as C
    from kintinuous_amd import abi
    p = _plane_cloud(5000, seed=9)
    ws = C.c_void_p()
    abi._chk(abi.lib().kt_slice_ws_create(ctx.h, 8192, None, C.byref(ws)))
    pts = ctx.upload(p)
    for count, bound in ((3000, 5000), (9999, 4000)):
        n_dev = ctx.upload(np.array([count], np.uint32))
        abi._chk(abi.lib().kt_slice_process_device(ws, pts.ptr, n_dev.ptr, bound, 100, 0.05, 20))
        n = C.c_size_t(0)
        abi._chk(abi.lib().kt_slice_ws_count(ws, C.byref(n)))
        want = oracle_mod.slice_process(p[: min(count, bound)], 100, 0.05)
        assert n.value == len(want) > 0
        got = np.zeros(n.value, abi.NPOINT_DTYPE)
        abi._chk(abi.lib().kt_download(ctx.h, got.ctypes.data_as(C.c_void_p), abi.lib().kt_slice_ws_output(ws), got.nbytes))
        assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["bgra"], want["bgra"])
    abi._chk(abi.lib().kt_slice_ws_destroy(ws))
