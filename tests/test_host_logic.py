"""Host-side logic that needs no GPU: synthetic data, .klg I/O, the frame schedule, and the oracle's tracker state machine
(shift decisions, slices, pose bookkeeping) on a small sequence."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_is_deterministic_and_sane():
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    R, c = synth.orbit_trajectory(4)[3]
    a = synth.render(synth.Scene("room"), cam, R, c)
    b = synth.render(synth.Scene("room"), cam, R, c)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    d, rgb = a
    assert d.dtype == np.uint16 and rgb.dtype == np.uint8 and d.shape == (120, 160) and rgb.shape == (120, 160, 3)
    assert (d > 0).all() and d.max() < 4000 and rgb.min() >= 1  # closed room: every ray hits; intensity never 0
    other = synth.render(synth.Scene("room", seed=1235), cam, R, c)
    assert not np.array_equal(other[0], d)  # per-stream seeds move the sphere / cube
    # per-frame motion stays inside the projective-ICP basin
    tr = synth.orbit_trajectory(300)
    step = max(np.linalg.norm(tr[k + 1][1] - tr[k][1]) for k in range(299))
    assert step < 0.02
    cw = synth.crabwalk_trajectory(420)
    assert abs(cw[200][1][0] - 3.0) < 1e-9 and abs(cw[400][1][0]) < 1e-9


def test_klg_roundtrip(tmp_path):
    from kintinuous_amd import klg, synth
    cam = synth.Camera.small(64, 48)
    tr = synth.orbit_trajectory(4)
    frames = [synth.render(synth.Scene("room"), cam, R, c) for (R, c) in tr]
    for comp in (False, True):
        p = str(tmp_path / f"s{int(comp)}.klg")
        klg.write_klg(p, frames, cols=64, rows=48, compress_depth=comp)
        got = list(klg.read_klg(p, 64, 48))
        assert len(got) == 3  # the reference reader never yields the last frame (RawLogReader.cpp:147-150)
        for k, (ts, d, rgb) in enumerate(got):
            assert ts == 33333 * k and np.array_equal(d, frames[k][0]) and np.array_equal(rgb, frames[k][1])
        assert len(list(klg.read_klg(p, 64, 48, reference_quirk=False))) == 4


def test_pingpong_schedule():
    from kintinuous_amd.multistream import pingpong, stream_seed
    assert [pingpong(i, 4) for i in range(10)] == [0, 1, 2, 3, 2, 1, 0, 1, 2, 3]
    assert all(abs(pingpong(i + 1, 7) - pingpong(i, 7)) == 1 for i in range(50))
    assert stream_seed(0) == 1234 and stream_seed(7) == 1241


def test_oracle_tracker_shifts_and_bookkeeping(oracle_mod):
    """KintinuousTracker state machine in the oracle: moving the camera +x past the shift threshold extracts a slab,
    clears it, advances voxelWrap and pulls the translation back; the global camera position stays continuous."""
    from kintinuous_amd import synth
    from oracle.oracle import OTrackerConfig, OracleTracker
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    idx = list(range(0, 30, 2))
    N, size, shift = 64, 7.0, 2
    trk = OracleTracker(OTrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, size, shift, 2, 0, 0, 0, 0, 0, 0))
    prev_g = None
    for k, i in enumerate(idx):
        d, rgb = synth.render(scene, cam, *traj[i])
        trk.process_frame(d, rgb, 1000 * k)
        R, t, g = trk.pose()
        if prev_g is not None:
            assert np.linalg.norm(g - prev_g) < 0.08  # continuous although t is pulled back on every shift
        prev_g = g
        cell = size / N
        assert np.all(np.abs(t - size / 2) < (shift + 1.5) * cell + 0.35)  # the camera stays near the volume centre
    w = trk.voxel_wrap()
    assert w[0] >= 2 and trk.num_slices() >= 1
    pts, dim = trk.slice(0)
    assert dim == 0  # XPlus (the first slabs leave the volume at its unobserved left edge, so they may be empty)
    assert trk.num_poses() == len(idx)
    ts, P, loop = trk.dense_pose(0)
    assert ts == 0 and loop and np.allclose(P[:3, :3], np.eye(3)) and np.allclose(P[:3, 3], 0)
    ts, P, loop = trk.dense_pose(len(idx) - 1)
    Rg, cg = traj[idx[-1]]
    assert not loop and np.abs(P[:3, 3] - cg).max() < 0.06  # global camera == scene coordinates (volume centred on camera 0)
    trk.finalise()
    pts, dim = trk.slice(trk.num_slices() - 1)
    assert dim == 7 and len(pts) > 300  # FINAL: the whole remaining surface
    assert pts["bgra"][:, 3].min() >= 1  # alpha carries the voxel weight (CloudSliceProcessor culls on it)
    trk.close()


def test_oracle_static_mode_never_shifts(oracle_mod):
    from kintinuous_amd import synth
    from oracle.oracle import OTrackerConfig, OracleTracker
    cam = synth.Camera.small(160, 120)
    trk = OracleTracker(OTrackerConfig(cam.cols, cam.rows, 48, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 1, 2, 1, 0, 0, 0, 0, 0))
    R, t, g = trk.pose()
    assert t[2] == pytest.approx(-0.45) and t[0] == pytest.approx(3.0)  # camera parked 0.45 m outside the near face (:101-110)
    for k, (Rk, ck) in enumerate(synth.static_trajectory(3)):
        d, rgb = synth.render(synth.Scene("farwall"), cam, Rk, ck)
        trk.process_frame(d, rgb, k)
    assert trk.num_slices() == 0 and not trk.voxel_wrap().any()
    assert abs(trk.trunc_dist() - 2.1 * 6.0 / 48) < 1e-6  # clamped to >= 2.1 voxels (TSDFVolume.cpp:89-97)
    trk.close()


def test_host_math_abi_matches_oracle(oracle_mod):
    """kt_host_* (the Gauss-Newton host math exported for per-operator callers) vs the oracle's restatement of
    Eigen LDLT / cv::Rodrigues / Matrix3f::inverse: bit-identical (both are the same published algorithms in double)."""
    import numpy as np
    from kintinuous_amd import abi
    rng = np.random.default_rng(7)
    # well conditioned, rank deficient, badly scaled and ICP-like (float sums widened) systems
    for trial in range(3000):
        J = rng.standard_normal((40, 6)) * rng.uniform(0.1, 10.0, 6)
        A = J.T @ J
        if trial % 5 == 4:
            A[:, 3] = A[:, 2]; A[3, :] = A[2, :]          # rank deficient: pseudo-inverse branch
        if trial % 7 == 3:
            A *= 10.0 ** rng.integers(-200, 200)
        if trial % 11 == 5 and trial % 7 != 3:
            A = A.astype(np.float32).astype(np.float64)   # float sums widened, as the trackers feed it
        b = rng.standard_normal(6) * 10.0 ** rng.integers(-3, 4)
        if trial % 13 == 0:
            b[rng.integers(0, 6)] = 0.0
        x, xo = abi.host_ldlt_solve6(A, b), oracle_mod.ldlt_solve6(A, b)
        assert np.array_equal(x.view(np.uint64), xo.view(np.uint64)), trial
    for trial in range(20):
        r = rng.standard_normal(3) * (1e-9 if trial == 0 else 0.3)
        assert np.array_equal(abi.host_rodrigues(r).view(np.uint64), oracle_mod.rodrigues(r).view(np.uint64))
        from tests.conftest import random_rotation
        R = random_rotation(rng, 1.0)
        assert np.array_equal(abi.host_mat33_inverse(R).view(np.uint32), oracle_mod.mat33_inverse(R).view(np.uint32))
    # pose update: identity increment keeps the pose; a pure translation increment moves the camera by -R*t
    rt, Rc, tc = abi.host_pose_update(np.zeros(6), np.eye(4), np.eye(3), [1, 2, 3])
    assert np.array_equal(Rc, np.eye(3, dtype=np.float32)) and np.array_equal(tc, np.array([1, 2, 3], np.float32))
    rt, Rc, tc = abi.host_pose_update([0.5, 0, 0, 0, 0, 0], np.eye(4), np.eye(3), [1, 2, 3])
    assert np.allclose(tc, [0.5, 2, 3]) and rt[0, 3] == 0.5


def test_reposition_cube_abi_matches_oracle(oracle_mod):
    """kt_host_reposition_cube vs the oracle over random rotations -- the general branch, the small-angle branch (identity and
    nearly identity), rotations by almost pi (axis from the diagonal) and not-quite-orthonormal matrices: bit-identical bases."""
    from kintinuous_amd import abi
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(11)
    cases = [np.eye(3)]
    for k in range(200):
        axis = rng.standard_normal(3)
        axis /= np.linalg.norm(axis)
        angle = [rng.uniform(0, np.pi), 1e-7 * rng.uniform(), np.pi - 1e-6 * rng.uniform(), np.pi][k % 4]
        R = Rotation.from_rotvec(axis * angle).as_matrix()
        if k % 7 == 0:
            R = R + rng.standard_normal((3, 3)) * 1e-4          # tracked rotations drift off SO(3) a little
        cases.append(R)
    moved = 0
    for R in cases:
        R = R.astype(np.float32)
        t = (np.array([3.0, 3.0, 0.0]) + rng.uniform(-0.6, 0.6, 3)).astype(np.float32)
        basis = np.array([3.0, 3.0, 0.0], np.float32)
        for thresh in (2, 14):
            a = abi.host_reposition_cube(R, t, 6.0, [6.0 / 96] * 3, thresh, basis)
            b = oracle_mod.reposition_cube(R, t, 6.0, [6.0 / 96] * 3, thresh, basis)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (R, t, thresh, a, b)
            moved += int(not np.array_equal(a, basis))
    assert moved > 50      # both outcomes are exercised


def test_rgbd_and_ground_truth_host_math_known_answers():
    """kt_host_compute_krk, kt_host_trajectory_pose, kt_host_ground_truth_pose (the host math of host/RGBDOdometry.h and
    host/GroundTruthOdometry.h) against float64 closed forms."""
    from kintinuous_amd import abi
    rng = np.random.default_rng(3)

    def rot(axis, a):
        axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K

    # K R K^-1, K t of the inverse of a rigid increment
    fx, fy, cx, cy = 525.0, 520.0, 319.5, 239.5
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    for _ in range(20):
        Rt = np.eye(4)
        Rt[:3, :3] = rot(rng.normal(size=3), rng.uniform(-0.2, 0.2))
        Rt[:3, 3] = rng.uniform(-0.1, 0.1, 3)
        inv = np.linalg.inv(Rt)
        krk, kt = abi.host_compute_krk(Rt, fx, fy, cx, cy)
        assert np.allclose(krk, K @ inv[:3, :3] @ np.linalg.inv(K), rtol=1e-6, atol=1e-4)
        assert np.allclose(kt, K @ inv[:3, 3], rtol=1e-6, atol=1e-5)
    krk, kt = abi.host_compute_krk(np.eye(4), fx, fy, cx, cy)
    assert np.array_equal(krk, np.eye(3, dtype=np.float32)) and not kt.any()

    # trajectory line -> pose: unit quaternion gives the rotation matrix, translation copied
    for _ in range(20):
        ax, a = rng.normal(size=3), rng.uniform(-3, 3)
        ax /= np.linalg.norm(ax)
        q = np.r_[np.sin(a / 2) * ax, np.cos(a / 2)]
        t = rng.uniform(-2, 2, 3)
        T = abi.host_trajectory_pose(np.r_[t, q])
        assert np.allclose(T[:9].reshape(3, 3), rot(ax, a), atol=2e-6) and np.array_equal(T[9:], t.astype(np.float32))

    # ground-truth increment: current = last * M^-1 * (A^-1 B) * M with M = camera axes -> volume axes (x, y, z) -> (z, -x, -y)
    M = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]], np.float64)
    def mat(T12):
        m = np.eye(4); m[:3, :3] = np.asarray(T12[:9], np.float64).reshape(3, 3); m[:3, 3] = T12[9:]; return m
    for _ in range(20):
        A = np.r_[rot(rng.normal(size=3), rng.uniform(-1, 1)).ravel(), rng.uniform(-1, 1, 3)].astype(np.float32)
        B = np.r_[rot(rng.normal(size=3), rng.uniform(-1, 1)).ravel(), rng.uniform(-1, 1, 3)].astype(np.float32)
        last = np.r_[rot(rng.normal(size=3), rng.uniform(-1, 1)).ravel(), rng.uniform(2, 4, 3)].astype(np.float32)
        R, t = abi.host_ground_truth_pose(A, B, last[:9], last[9:])
        want = mat(last) @ np.linalg.inv(M) @ np.linalg.inv(mat(A)) @ mat(B) @ M
        assert np.allclose(R, want[:3, :3], atol=5e-6) and np.allclose(t, want[:3, 3], atol=1e-5)
    # no motion between equal stamps
    R, t = abi.host_ground_truth_pose(A, A, last[:9], last[9:])
    assert np.allclose(R, last[:9].reshape(3, 3), atol=1e-6) and np.allclose(t, last[9:], atol=1e-6)



def test_eigen_adapters_type_check(tmp_path):
    """host/EigenAdapters.h (the conversions to the Eigen / PCL types the reference's backend passes around, KintinuousTracker.h:59-80)
    sits behind __has_include and had never been through a compiler: this image has neither library.  tests/stubs/ holds stand-in
    headers with the SHAPES of the types it touches (fixed-size matrices with a storage order, Map, pcl::PointCloud, the two point
    layouts); the adapters are compiled against them and every one is run once (header-only, no GPU)."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "eigen_adapters_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "kintinuous_amd", "host"),
                        "-I", ROOT, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "stubs", "eigen_adapters_check.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "eigen adapters ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_thread_object_protocol(tmp_path):
    """host/ThreadObject.h, ThreadMutexObject.h and ThreadDataPack.h without a GPU: start / running / stop / restart, a loop that ends by
    itself, a consumer woken by the tracker's signal, the end-of-run hand-shake (tests/stubs/thread_object_check.cpp).  The shell's
    headers call into the C-ABI, so the program links libkt_hip.so -- it never creates a context."""
    import subprocess
    from kintinuous_amd import build
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(build.OUT):
        pytest.skip("libkt_hip.so not built")
    exe = str(tmp_path / "thread_object_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "kintinuous_amd", "host"), "-I", ROOT,
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "stubs", "thread_object_check.cpp"), "-o", exe,
                        "-L", os.path.dirname(build.OUT), "-lkt_hip", "-lz", "-Wl,-rpath," + os.path.dirname(build.OUT), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "thread object ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("kt_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_gpus_flag_resolves_the_launch():
    """`python bench.py --gpus N` means N ranks in every launch form (the flag was parsed and never read): bare -> re-exec under
    torch.distributed.run; under a launcher -> WORLD_SIZE must agree; no flag -> the launcher's world, else one rank."""
    b = _bench_module()
    assert b.resolve_launch(None, {}) == ("run", 0, 0, 1)
    assert b.resolve_launch(1, {}) == ("run", 0, 0, 1)
    assert b.resolve_launch(8, {}) == ("exec", 8)
    env = {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2"}
    assert b.resolve_launch(4, env) == ("run", 2, 2, 4) and b.resolve_launch(None, env) == ("run", 2, 2, 4)
    with pytest.raises(SystemExit):
        b.resolve_launch(8, env)
    with pytest.raises(SystemExit):
        b.resolve_launch(1, env)
    with pytest.raises(SystemExit):
        b.resolve_launch(0, {})


@pytest.mark.parametrize("form", ["bare", "torchrun"])
def test_bench_gpus_2_starts_two_ranks(form, tmp_path):
    """Both launch forms of `bench.py --gpus 2` end up as a two-rank job whose ranks meet in the key-value store (--launch-check stops there:
    no GPU is touched); rank 0's stdout is ONE JSON line with n_gpus == --gpus."""
    import json
    import socket
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    if form == "bare":
        cmd = [sys.executable, bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-check"]
    else:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-check"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 == j["gpus_flag"] and j["highest_rank_seen"] == 1


def test_squared_thresholds_of_the_icp_row_are_equivalent_to_the_roots():
    """csrc/kt_track.hip compares the SQUARES dist^2 / sine^2 with thresholds found once on the host instead of taking two correctly rounded
    square roots per pixel (reduce.cu:247-253: `sine < angleThres && dist <= distThres`).  Equivalence, checked where it can fail -- at the
    threshold's neighbours: for X = kt_debug_sq_threshold(T, strict), every float x within 64 ulps of X satisfies
    (sqrtf(x) <= T) == (x <= X)   [strict: (sqrtf(x) < T) == (x <= X)], and the corner cases (0, denormals, huge, inf, NaN, negative)."""
    import ctypes as C
    from kintinuous_amd import abi, build
    build.build()
    l = C.CDLL(abi.LIB_PATH)
    f = l.kt_debug_sq_threshold
    f.restype, f.argtypes = C.c_float, [C.c_float, C.c_int]
    rng = np.random.default_rng(11)
    Ts = np.concatenate([np.float32([0.10, 0.34202015, 0.05, 1.0, 2.0, 1e-3, 3.0e-20, 1e19, 0.5, 0.70710677, 1.1754944e-38, 1e-45, 3.4e38]),
                         (10.0 ** rng.uniform(-18, 18, 400)).astype(np.float32), rng.uniform(0.0, 1.0, 400).astype(np.float32)])
    for T in Ts.tolist():
        T = float(np.float32(T))
        for strict in (0, 1):
            X = np.float32(f(T, strict))
            assert X >= 0
            bits = int(X.view(np.uint32))
            lo, hi = max(0, bits - 64), min(0x7f7fffff, bits + 64)
            xs = np.arange(lo, hi + 1, dtype=np.uint32).view(np.float32)
            roots = np.sqrt(xs)                                  # numpy's float32 sqrt is the correctly rounded IEEE root
            want = (roots < np.float32(T)) if strict else (roots <= np.float32(T))
            assert np.array_equal(want, xs <= X), (T, strict, X)
    assert f(0.0, 0) == 0.0 and f(0.0, 1) == -1.0                # sqrtf(x) <= 0 only at 0; sqrtf(x) < 0 never
    assert f(-1.0, 0) == -1.0 and f(float("nan"), 0) == -1.0 and f(float("nan"), 1) == -1.0
    assert f(float("inf"), 0) == float("inf") and np.float32(f(float("inf"), 1)) == np.finfo(np.float32).max


def test_kernel_stats_splitter_moves_the_counting_rows(tmp_path):
    """scripts/split_kernel_stats.py: the counting variants of bench.py's untimed replay (`<true, ...>` template rows) go below a separator row and
    out of the percentages, the frame's own launches keep their order and sum to 100 %."""
    import csv
    import subprocess
    import sys
    src, dst = tmp_path / "in.csv", tmp_path / "out.csv"
    rows = [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"],
            ["void kt_raycast_kernel<true, true, true>(kt_raycast_args)", "10", "2600000", "260000", "50.0", "1", "2", "0"],
            ["kt_icp_level_kernel(kt_icp_args)", "30", "1500000", "50000", "28.8", "1", "2", "0"],
            ["void kt_tsdf23_lean_kernel<false, false, true>(kt_tsdf_lean_args)", "10", "500000", "50000", "9.6", "1", "2", "0"],
            ["void kt_tsdf23_lean_kernel<true, false, true>(kt_tsdf_lean_args)", "10", "600000", "60000", "11.5", "1", "2", "0"]]
    with open(src, "w", newline="") as f:
        csv.writer(f).writerows(rows)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "split_kernel_stats.py"), str(src), str(dst)])
    out = list(csv.reader(open(dst)))
    names = [r[0] for r in out[1:]]
    assert names[0].startswith("kt_icp_level_kernel") and names[1].startswith("void kt_tsdf23_lean_kernel<false")
    assert names[2].startswith("# BELOW") and all("<true," in n for n in names[3:]) and len(names) == 5
    assert abs(sum(float(r[4]) for r in out[1:3]) - 100.0) < 1e-3 and all(r[4] == "" for r in out[4:])
