"""f1: what CloudSliceProcessor::save writes (backend/CloudSliceProcessor.cpp:180-231) -- the final pcl::VoxelGrid<PointXYZRGBNormal>
and the binary PCD.  CPU tests hold the library's host code (kt_host_voxel_grid_normal / kt_host_save_pcd, no GPU involved) against the
oracle's restatement and against a plain numpy model of the published algorithm; the GPU test runs the C++ driver's `-pcd` and compares
the file with the oracle's pipeline on the same slices."""
import os
import subprocess

import numpy as np
import pytest

from kintinuous_amd import abi, klg


def _random_processed_cloud(rng, n, extent=0.5, with_nan=True):
    pts = np.zeros(n, abi.NPOINT_DTYPE)
    pts["xyz"] = rng.uniform(-extent, extent, (n, 3)).astype(np.float32) + np.float32([1.0, -2.0, 3.0])
    pts["one"] = 1.0
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    pts["normal"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    pts["bgra"] = rng.integers(0, 256, (n, 4)).astype(np.uint8)
    pts["bgra"][:, 3] = 0     # a processed slice carries a zero alpha byte
    pts["curvature"] = rng.uniform(0, 0.3, n).astype(np.float32)
    if with_nan and n > 10:   # NormalEstimation leaves NaN normals on points with fewer than 3 neighbours
        k = rng.integers(0, n, 3)
        pts["normal"][k] = np.nan
        pts["curvature"][k] = np.nan
    return pts


def _model_voxel_grid(pts, leaf):
    """voxel_grid.hpp applyFilter, downsample_all_data_, PointXYZRGBNormal: sequential float32 arithmetic, leaves in key order, the
    points of a leaf in input order, `centroid /= n` as Eigen 3.2's multiplication by 1 / n."""
    f = np.float32
    inv = f(1.0) / f(leaf)
    xyz = pts["xyz"]
    mn, mx = xyz.min(axis=0), xyz.max(axis=0)
    min_b = np.floor(mn * inv).astype(np.int64)
    div_b = np.floor(mx * inv).astype(np.int64) - min_b + 1
    ijk = (np.floor(xyz * inv) - min_b.astype(np.float32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div_b[0] + ijk[:, 2] * div_b[0] * div_b[1]
    order = np.argsort(key, kind="stable")
    out = []
    i = 0
    while i < len(order):
        j = i
        acc = None
        while j < len(order) and key[order[j]] == key[order[i]]:
            p = pts[order[j]]
            t = np.array([p["xyz"][0], p["xyz"][1], p["xyz"][2], p["bgra"].view(np.float32)[0], p["normal"][0], p["normal"][1], p["normal"][2],
                          p["curvature"], f(p["bgra"][2]), f(p["bgra"][1]), f(p["bgra"][0])], np.float32)
            with np.errstate(all="ignore"):
                acc = t.copy() if acc is None else (acc + t).astype(np.float32)
            j += 1
        with np.errstate(all="ignore"):
            acc = (acc * (f(1.0) / f(j - i))).astype(np.float32)
        o = np.zeros((), abi.NPOINT_DTYPE)
        o["xyz"] = acc[:3]
        o["one"] = 1.0
        o["normal"] = acc[4:7]
        o["curvature"] = acc[7]
        o["bgra"] = [int(acc[10]), int(acc[9]), int(acc[8]), 0]
        out.append(o)
        i = j
    return np.array(out, abi.NPOINT_DTYPE)


def _same_points(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("seed", range(4))
def test_final_voxel_grid_matches_oracle_and_model(oracle_mod, seed):
    rng = np.random.default_rng(400 + seed)
    n = [2000, 5000, 300, 1][seed]
    leaf = [0.05, 0.0234375, 0.2, 0.01][seed]
    pts = _random_processed_cloud(rng, n)
    got = abi.voxel_grid_normal(pts, leaf)
    want = oracle_mod.voxel_grid_normal(pts.view(oracle_mod.NPOINT_DTYPE), leaf)
    assert 0 < len(got) <= n and (n < 100 or len(got) < n)      # several points per leaf
    assert _same_points(got, want.view(abi.NPOINT_DTYPE))
    assert _same_points(got, _model_voxel_grid(pts, leaf))
    assert not got["bgra"][:, 3].any()                           # the re-packed rgb has a zero alpha byte


def test_final_voxel_grid_edge_cases(oracle_mod):
    empty = np.zeros(0, abi.NPOINT_DTYPE)
    assert len(abi.voxel_grid_normal(empty, 0.01)) == 0 and len(oracle_mod.voxel_grid_normal(empty.view(oracle_mod.NPOINT_DTYPE), 0.01)) == 0
    # "Leaf size is too small for the input dataset": more than 2^31 leaves in the bounding box -> the cloud passes through untouched
    rng = np.random.default_rng(7)
    pts = _random_processed_cloud(rng, 50, extent=40.0, with_nan=False)
    pts["bgra"][:, 3] = 9
    got = abi.voxel_grid_normal(pts, 0.01)
    assert _same_points(got, pts) and _same_points(oracle_mod.voxel_grid_normal(pts.view(oracle_mod.NPOINT_DTYPE), 0.01).view(abi.NPOINT_DTYPE), pts)
    # all points in one leaf: one output point, the means of every field
    one = _random_processed_cloud(rng, 7, extent=0.001, with_nan=False)
    one["xyz"] += np.float32(0.5)     # (the cloud's centre (1, -2, 3) sits on leaf faces)
    got = abi.voxel_grid_normal(one, 1.0)
    assert len(got) == 1 and _same_points(got, _model_voxel_grid(one, 1.0))


@pytest.mark.parametrize("n", [0, 1, 1234])
def test_pcd_writer_matches_oracle(oracle_mod, tmp_path, n):
    rng = np.random.default_rng(n)
    pts = _random_processed_cloud(rng, n)
    path = str(tmp_path / "cloud.pcd")
    abi.save_pcd(path, pts)
    data = open(path, "rb").read()
    assert data == oracle_mod.pcd_binary(pts.view(oracle_mod.NPOINT_DTYPE))
    header, payload = data.split(b"DATA binary\n", 1)
    assert header.decode("ascii").splitlines() == [
        "# .PCD v0.7 - Point Cloud Data file format", "VERSION 0.7", "FIELDS x y z rgb normal_x normal_y normal_z curvature",
        "SIZE 4 4 4 4 4 4 4 4", "TYPE F F F F F F F F", "COUNT 1 1 1 1 1 1 1 1", "WIDTH %d" % n, "HEIGHT 1", "VIEWPOINT 0 0 0 1 0 0 0",
        "POINTS %d" % n]
    assert len(payload) == 32 * n
    back = klg.read_pcd(path)
    assert np.array_equal(back["xyz"], pts["xyz"]) and np.array_equal(back["bgra"], pts["bgra"])
    assert back["normal"].tobytes() == pts["normal"].tobytes() and back["curvature"].tobytes() == pts["curvature"].tobytes()


def test_pcd_writer_reports_unwritable_path(tmp_path):
    with pytest.raises(Exception):
        abi.save_pcd(str(tmp_path / "no_such_dir" / "x.pcd"), np.zeros(1, abi.NPOINT_DTYPE))


@pytest.mark.gpu
@pytest.mark.parametrize("nos", [False, True])
def test_driver_pcd_matches_the_oracle_pipeline(ctx, oracle_mod, tmp_path, nos):
    """`kintinuous_hip -pcd [-nos]` on a shifting log: the slice-processor thread behind the tracker + save().  The same slices (taken
    from the Python binding of the same tracker) go through the oracle: kto_slice_process per slice, concatenation over slices
    [1, latestPoseId), the final VoxelGrid with -nos, kto_pcd_binary.  Header and point count must be equal; positions and colours are
    compared leaf by leaf (extraction order inside a slice is free, so a leaf's float sum may differ in its last bits between two runs),
    normals within the tolerance of tests/test_slice_process.py."""
    from kintinuous_amd import build, synth
    build.build_host()
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    idx = list(range(0, 40, 2)) + list(range(40, 0, -2))
    frames = [synth.render(scene, cam, *traj[i]) for i in idx]
    log = str(tmp_path / "seq.klg")
    klg.write_klg(log, list(frames) + [frames[-1]], cols=cam.cols, rows=cam.rows)   # (the last frame of a log is never processed)
    calib = str(tmp_path / "calib.txt")
    with open(calib, "w") as f:
        f.write(f"{cam.fx!r} {cam.fy!r} {cam.cx!r} {cam.cy!r}\n")
    N, size, cw = 96, 7.0, 8
    cmd = [build.HOST_BIN, "-l", log, "-c", calib, "-n", str(N), "-w", str(cam.cols), "-h", str(cam.rows), "-s", str(size), "-t", "3", "-o",
           str(tmp_path / "out"), "-pcd"] + (["-nos"] if nos else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PCD saved" in r.stdout, r.stdout + r.stderr
    got = klg.read_pcd(str(tmp_path / "out.pcd"))

    cfg = abi.TrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, size, 3, 2, 0, 0, 0, 0, 0, 0)
    trk = abi.Tracker(ctx, cfg)
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame_host(d, rgb, 33333 * (k + 1))
    trk.finalise()
    leaf = np.float32(size) / np.float32(N)
    parts = []
    for i in range(trk.num_slices()):
        pts, _dim = trk.slice(i)
        parts.append(oracle_mod.slice_process(pts.view(oracle_mod.POINT_DTYPE), cw, float(leaf)))
    trk.close()
    full = np.concatenate(parts) if parts else np.zeros(0, oracle_mod.NPOINT_DTYPE)
    assert len(full) > 1000
    if nos:
        full = oracle_mod.voxel_grid_normal(full, float(leaf))
    want_bytes = oracle_mod.pcd_binary(full)
    data = open(tmp_path / "out.pcd", "rb").read()
    assert data.split(b"DATA binary\n", 1)[0] == want_bytes.split(b"DATA binary\n", 1)[0]      # header incl. the point count
    assert len(got) == len(full)
    assert np.abs(got["xyz"] - full["xyz"]).max() <= 2e-6 * max(1.0, float(np.abs(full["xyz"]).max()))
    assert np.abs(got["bgra"].astype(int) - full["bgra"].astype(int)).max() <= 1
    ok = np.isfinite(full["normal"]).all(axis=1) & np.isfinite(got["normal"]).all(axis=1)
    assert ok.mean() > 0.95
    close = np.abs(got["normal"][ok] - full["normal"][ok]).max(axis=1) <= 2e-3
    assert close.mean() > 0.995, close.mean()
