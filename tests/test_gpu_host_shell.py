"""GPU: the C++ host shell (kintinuous_amd/host: internal.h wrappers, DeviceArray, TsdfVolume, ICPOdometry, KintinuousTracker,
TrackerInterface, RawLogReader) driven through its headless binary on a synthetic .klg log.
Checks: (1) the operator path (every frame composed on the host from the reference-named operators, host Gauss-Newton loop)
and the device-resident path write byte-identical .poses files; (2) those poses equal the Python C-ABI tracker's and the
oracle's; (3) shifting logs produce the same slices on both paths."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kintinuous_amd", "host", "bin", "kintinuous_hip")


def _run(args, cwd):
    env = dict(os.environ)
    r = subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def _controller(args, cwd):
    exe = os.path.join(ROOT, "kintinuous_amd", "host", "bin", "controller_test")
    assert os.path.exists(exe), "controller_test not built: run __graft_entry__.build()"
    r = subprocess.run([exe] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("controller ")]
    assert r.returncode == 0 and line, r.stdout + r.stderr
    tok = line[-1].split()
    got = {k: tok[tok.index(k) + 1] for k in ("frames", "slices", "consumed", "saved", "finished")}
    got["fps"] = float(tok[tok.index("frames/s") - 1])      # by the child's own steady clock
    return got, r.stdout


def _announced(out, *words):
    """ThreadObject announcements ("<identifier> started" / "<identifier> ended") as WHOLE LINES of the child's stdout: each one leaves
    the process as a single locked stdio write (host/ThreadObject.h: announce), so concurrent threads cannot interleave inside a line."""
    lines = set(l.strip() for l in out.splitlines())
    return all(w in lines for w in words)



def _summary(out):
    tok = [l for l in out.splitlines() if l.startswith("frames ")][-1].split()
    return {"frames": int(tok[1]), "slices": int(tok[3]), "points": int(tok[5])}


def _make_log(tmp_path, cam, frames, name="log.klg", zlib_depth=False):
    from kintinuous_amd import klg
    path = str(tmp_path / name)
    # the reader never returns the last frame of a log (reference quirk): append a sentinel
    klg.write_klg(path, list(frames) + [frames[-1]], cols=cam.cols, rows=cam.rows, compress_depth=zlib_depth)
    calib = str(tmp_path / "calib.txt")
    with open(calib, "w") as f:
        f.write(f"{cam.fx!r} {cam.fy!r} {cam.cx!r} {cam.cy!r}\n")
    return path, calib


def _poses(path):
    return np.loadtxt(path, ndmin=2)


def test_binary_exists():
    assert os.path.exists(BIN), "host shell not built: run __graft_entry__.build()"


def test_operator_path_equals_device_path(ctx, oracle_mod, small_scene, tmp_path):
    from kintinuous_amd import abi
    cam, frames, traj = small_scene
    frames = frames[:6]
    log, calib = _make_log(tmp_path, cam, frames, zlib_depth=True)
    common = ["-l", log, "-c", calib, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "6"]
    a = _summary(_run(common + ["-o", str(tmp_path / "dev"), "-ppm"], str(tmp_path)))
    b = _summary(_run(common + ["-o", str(tmp_path / "ops"), "-ops", "-ppm"], str(tmp_path)))
    assert a == b and a["frames"] == len(frames) and a["slices"] == 1  # FINAL slice only
    # -ppm: getImage / getModelDepth views of the final model, identical on both paths and not blank
    for name in ("_model.ppm", "_color.ppm", "_depth.pgm"):
        va, vb = open(str(tmp_path / "dev") + name, "rb").read(), open(str(tmp_path / "ops") + name, "rb").read()
        assert va == vb and len(va) > cam.cols * cam.rows
    header = f"P6\n{cam.cols} {cam.rows}\n255\n".encode()
    model = np.frombuffer(open(str(tmp_path / "dev_model.ppm"), "rb").read()[len(header):], np.uint8).reshape(cam.rows, cam.cols, 3)
    assert (model.max(axis=2) >= 20).mean() > 0.5       # shaded pixels carry the +20 ambient term
    ta, tb = open(tmp_path / "dev.poses").read(), open(tmp_path / "ops.poses").read()
    assert ta == tb and len(ta.splitlines()) == len(frames) - 1

    # same poses as the Python binding of the same C-ABI tracker (and hence as the oracle, test_gpu_tracker.py)
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
    trk = abi.Tracker(ctx, cfg)
    P = _poses(tmp_path / "dev.poses")
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame_host(d, rgb, 33333 * k)
        if k >= 1:
            _, _, gc = trk.pose()
            assert np.allclose(P[k - 1, 1:4], gc, rtol=2e-6, atol=1e-6)      # %g keeps 6 significant digits
            assert abs(P[k - 1, 0] - 33333 * k / 1e6) < 1e-6
    trk.close()


@pytest.mark.parametrize("flags", [["-r"], ["-ri"], ["-r", "-dc"], ["-ri", "-fod"]])
def test_operator_path_rgbd_odometry(small_scene, tmp_path, flags):
    """host/RGBDOdometry.h (the reference's RGBDOdometry composed from computeRgbResidual / icpStep / rgbStep with the Gauss-Newton step
    on the host) against the device-resident tracker: byte-identical pose files and final views.  -dc exercises the reference's
    "no depth pyramid for pure RGB-D without the colour angle weight" branch (KintinuousTracker.cpp:465)."""
    cam, frames, traj = small_scene
    frames = frames[:5]
    log, calib = _make_log(tmp_path, cam, frames)
    common = ["-l", log, "-c", calib, "-n", "64", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "6"] + flags
    a = _summary(_run(common + ["-o", str(tmp_path / "dev"), "-ppm"], str(tmp_path)))
    b = _summary(_run(common + ["-o", str(tmp_path / "ops"), "-ops", "-ppm"], str(tmp_path)))
    assert a == b and a["frames"] == len(frames)
    ta, tb = open(tmp_path / "dev.poses").read(), open(tmp_path / "ops.poses").read()
    assert ta == tb and len(ta.splitlines()) == len(frames) - 1
    for name in ("_model.ppm", "_color.ppm", "_depth.pgm"):
        assert open(str(tmp_path / "dev") + name, "rb").read() == open(str(tmp_path / "ops") + name, "rb").read()
    # the camera moved: RGB-D odometry tracked something
    P = _poses(tmp_path / "dev.poses")
    assert np.abs(P[-1, 1:4] - P[0, 1:4]).max() > 1e-3


def test_operator_path_ground_truth_odometry(tmp_path):
    """host/GroundTruthOdometry.h on the operator path (-ops -p): same pose file as the device-resident tracker, dropped frame included."""
    from kintinuous_amd import klg, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("room")
    poses = synth.orbit_trajectory(8)
    frames = [synth.render(scene, cam, R, c) for (R, c) in poses]
    stamps = [33333 * (k + 1) for k in range(len(frames))]
    rows = synth.ground_truth_rows(poses)
    keep = [k for k in range(len(frames)) if k != 4]
    log = str(tmp_path / "gt.klg")
    klg.write_klg(log, list(frames) + [frames[-1]], timestamps=stamps + [stamps[-1] + 33333], cols=cam.cols, rows=cam.rows)
    calib = str(tmp_path / "calib.txt")
    with open(calib, "w") as f:
        f.write(f"{cam.fx!r} {cam.fy!r} {cam.cx!r} {cam.cy!r}\n")
    tfile = str(tmp_path / "traj.csv")
    synth.write_trajectory_file(tfile, [stamps[k] for k in keep], rows[keep])
    common = ["-l", log, "-c", calib, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "6", "-p", tfile]
    a = _summary(_run(common + ["-o", str(tmp_path / "dev"), "-ppm"], str(tmp_path)))
    b = _summary(_run(common + ["-o", str(tmp_path / "ops"), "-ops", "-ppm"], str(tmp_path)))
    assert a == b
    ta, tb = open(tmp_path / "dev.poses").read(), open(tmp_path / "ops.poses").read()
    assert ta == tb and len(ta.splitlines()) == len(keep) - 1
    for name in ("_model.ppm", "_color.ppm", "_depth.pgm"):
        assert open(str(tmp_path / "dev") + name, "rb").read() == open(str(tmp_path / "ops") + name, "rb").read()


def test_shifting_log_slices(tmp_path):
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    idx = list(range(0, 40, 2)) + list(range(40, 0, -2))  # 30 mm steps out and back (as test_gpu_tracker.test_shifting_crabwalk)
    frames = [synth.render(scene, cam, *traj[i]) for i in idx]
    log, calib = _make_log(tmp_path, cam, frames)
    common = ["-l", log, "-c", calib, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "7", "-t", "3"]
    a = _summary(_run(common + ["-o", str(tmp_path / "dev"), "-pcdraw"], str(tmp_path)))
    b = _summary(_run(common + ["-o", str(tmp_path / "ops"), "-ops", "-pcdraw"], str(tmp_path)))
    assert a == b
    assert a["slices"] >= 5 and a["points"] > 0      # X+ / X- shifts + FINAL
    assert open(tmp_path / "dev.poses").read() == open(tmp_path / "ops.poses").read()
    # -pcdraw: every slice point, x y z rgb; both paths extract the same point set (the order inside a slice is free)
    from kintinuous_amd import klg
    pa, pb = klg.read_pcd(str(tmp_path / "dev.raw.pcd")), klg.read_pcd(str(tmp_path / "ops.raw.pcd"))
    assert len(pa) == a["points"] == len(pb)
    key = lambda p: np.sort(np.ascontiguousarray(p).view(np.dtype((np.void, p.dtype.itemsize))).ravel())
    assert np.array_equal(key(pa), key(pb))
    assert np.isfinite(pa["xyz"]).all() and pa["bgra"][:, :3].any()


def test_ground_truth_trajectory_cli(ctx, tmp_path):
    """-p <trajectory>: the driver parses the reference's trajectory format, tracks with ground-truth poses, drops frames that have no
    entry, and writes the same poses as the Python binding of the same tracker."""
    from kintinuous_amd import abi, klg, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("room")
    poses = synth.orbit_trajectory(8)
    frames = [synth.render(scene, cam, R, c) for (R, c) in poses]
    stamps = [33333 * (k + 1) for k in range(len(frames))]
    rows = synth.ground_truth_rows(poses)
    keep = [k for k in range(len(frames)) if k != 4]
    log = str(tmp_path / "gt.klg")
    klg.write_klg(log, list(frames) + [frames[-1]], timestamps=stamps + [stamps[-1] + 33333], cols=cam.cols, rows=cam.rows)
    calib = str(tmp_path / "calib.txt")
    with open(calib, "w") as f:
        f.write(f"{cam.fx!r} {cam.fy!r} {cam.cx!r} {cam.cy!r}\n")
    tfile = str(tmp_path / "traj.csv")
    synth.write_trajectory_file(tfile, [stamps[k] for k in keep], rows[keep])
    out = _summary(_run(["-l", log, "-c", calib, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "6", "-p", tfile, "-o",
                         str(tmp_path / "gt")], str(tmp_path)))
    assert out["slices"] == 1
    P = _poses(tmp_path / "gt.poses")
    assert len(P) == len(keep) - 1                     # the first frame writes no line, the dropped frame none either
    # MainController::setup calls trackerInterface->loadTrajectory once more on top of the constructor's (MainController.cpp:112-116):
    # the controller stand-in does, and must write the same file
    gotc, outc = _controller(["-l", log, "-c", calib, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "6", "-p", tfile, "-o",
                              str(tmp_path / "gtc")], str(tmp_path))
    assert "Load trajectory:" in outc and gotc["finished"] == "1"
    assert open(str(tmp_path / "gt.poses"), "rb").read() == open(str(tmp_path / "gtc.poses"), "rb").read()

    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
    trk = abi.Tracker(ctx, cfg)
    trk.load_trajectory(np.array([stamps[k] for k in keep], np.uint64), rows[keep])
    line = 0
    for k, (d, rgb) in enumerate(frames):
        trk.process_frame_host(d, rgb, stamps[k])
        if k in keep and k > 0:
            _, _, gc = trk.pose()
            assert np.allclose(P[line, 1:4], gc, rtol=2e-6, atol=1e-6)
            assert abs(P[line, 0] - stamps[k] / 1e6) < 1e-6
            line += 1
    # ground truth keeps the camera on the rendered orbit
    R, t, _ = trk.pose()
    assert np.abs(t - (poses[-1][1] - poses[0][1] + 3.0)).max() < 1e-4
    trk.close()


def test_logger2_style_log_with_jpeg_colour(ctx, tmp_path):
    """zlib depth + JPEG colour (what the reference's logger records): the C++ reader decodes the colour with JpegDecoder.h; the
    RGB-D + ICP odometry then sees the same bytes as a Python run on frames decoded by the numpy reference decoder."""
    from kintinuous_amd import abi, klg, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("room")
    frames = [synth.render(scene, cam, R, c) for (R, c) in synth.orbit_trajectory(6)]
    log = str(tmp_path / "jpeg.klg")
    klg.write_klg(log, list(frames) + [frames[-1]], cols=cam.cols, rows=cam.rows, compress_depth=True, jpeg_quality=92)
    calib = str(tmp_path / "calib.txt")
    with open(calib, "w") as f:
        f.write(f"{cam.fx!r} {cam.fy!r} {cam.cx!r} {cam.cy!r}\n")
    out = _summary(_run(["-l", log, "-c", calib, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "6", "-ri", "-o", str(tmp_path / "j")],
                        str(tmp_path)))
    assert out["frames"] == len(frames)
    P = _poses(tmp_path / "j.poses")
    decoded = list(klg.read_klg(log, cols=cam.cols, rows=cam.rows))
    assert len(decoded) == len(frames)
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 1, 0, 0, 0)
    trk = abi.Tracker(ctx, cfg)
    for k, (ts, d, rgb) in enumerate(decoded):
        assert np.array_equal(d, frames[k][0])
        assert not np.array_equal(rgb, frames[k][1]) and np.abs(rgb.astype(int) - frames[k][1].astype(int)).mean() < 4   # lossy, but close
        trk.process_frame_host(d, rgb, ts)
        if k >= 1:
            _, _, gc = trk.pose()
            assert np.allclose(P[k - 1, 1:4], gc, rtol=2e-6, atol=1e-6)
    trk.close()


def test_backend_consumer_runs_against_the_shell(ctx, tmp_path):
    """The tracker-facing half of the reference's CloudSliceProcessor::process (backend/CloudSliceProcessor.cpp:38-83, 163-181), restated
    in host/consumer_test.cpp, compiles against host/KintinuousTracker.h and runs in its own thread while the tracker plays a shifting
    log: cloudMutex / cloudSignal / cycledMutex, init_utime, placeRecognitionBuffer[0], the 12-argument CloudSlice constructor,
    processedCloud, FINAL.  With -v the place-recognition tap fills placeRecognitionBuffer; the counts must match the C-ABI tracker's."""
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    frames = [synth.render(scene, cam, *traj[i]) for i in range(60)]
    log, calib = _make_log(tmp_path, cam, frames)
    exe = os.path.join(ROOT, "kintinuous_amd", "host", "bin", "consumer_test")
    assert os.path.exists(exe)
    args = ["-l", log, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "7", "-t", "3", "-v", "vocab.yml.gz", "-cw", "2"]
    r = subprocess.run([exe] + args, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    head = [l for l in r.stdout.splitlines() if l.startswith("consumed ")][0].split()   # (the ThreadObjects announce themselves first)
    assert _announced(r.stdout, "TrackerInterfaceThread started", "CloudSliceProcessorThread started", "TrackerInterfaceThread ended",
                      "CloudSliceProcessorThread ended"), r.stdout
    got = {k: head[head.index(k) + 1] for k in ("consumed", "finished", "first_utime", "pr", "loops", "poses", "latest", "first_frame")}
    # the same run through the C-ABI tracker.  consumer_test derives its intrinsics from the image size like MainController's default
    k = (528.0 * cam.cols / 640.0, 528.0 * cam.rows / 480.0, 320.0 * cam.cols / 640.0, 240.0 * cam.rows / 480.0)
    trk = abi.Tracker(ctx, abi.TrackerConfig(cam.cols, cam.rows, 96, k[0], k[1], k[2], k[3], 7.0, 3, 2, 0, 0, 0, 0, 0, 0, 0, 1))
    for i, (d, rgb) in enumerate(frames):
        trk.process_frame_host(d, rgb, 33333 * i)
    trk.finalise()
    nslices = trk.num_slices()
    assert nslices >= 4
    assert int(got["consumed"]) == nslices + 1 and got["finished"] == "1"            # + the consumer's own FIRST slice
    assert int(got["poses"]) == trk.num_poses() == int(got["latest"]) == len(frames)
    assert int(got["loops"]) == sum(trk.dense_pose(i)[2] for i in range(trk.num_poses()))
    assert int(got["pr"]) == len(trk.pr_samples()) >= 3
    assert int(got["first_utime"]) == 0 and got["first_frame"] == "1"
    live = head[head.index("live") + 1: head.index("live") + 3]
    assert int(live[0]) > 1000 and live[1] == "1"                                       # getLiveTsdf / getLiveImage were served
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("slice ")]
    assert len(lines) == nslices
    for i, l in enumerate(lines):
        n, dim = trk.slice_info(i)
        assert int(l[3]) == dim and int(l[9]) == trk.slice_pr_id(i)
        # processedCloud = kt_slice_process(cloud): exactly what the Python binding computes from the same slice; and, like the
        # reference (CloudSliceProcessor.cpp:112-140), the processor leaves the culled, voxel-gridded points in slice->cloud itself
        assert int(l[7]) == len(abi.slice_process(ctx, trk.slice(i)[0], 2, 7.0 / 96))
        assert int(l[5]) == int(l[7]) <= n
        assert (int(l[7]) > 0) == (n > 0) or int(l[7]) == 0
    trk.close()


def test_main_controller_runs_against_the_shell(tmp_path):
    """host/controller_test.cpp is MainController::setup / mainLoop / complete / save (MainController.cpp:93-183, 235-265) minus the GUI,
    compiled against the shell's TrackerInterface and CloudSliceProcessor AS ThreadObjects (start / stop / running, endRequested,
    loadTrajectory, the pauseCapture / finalised / cloudSliceProcessorFinished hand-shake, the limit throttle).  It must end by itself at
    the end of the log, write the same .poses as kintinuous_hip and a .pcd with the same points; -end K (the GUI's End button) must stop
    the run early through endRequested and still finalise and save; -limit must hold the tracker to 30 frames per second."""
    from kintinuous_amd import klg, synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    frames = [synth.render(scene, cam, *traj[i]) for i in range(0, 48, 2)]
    log, calib = _make_log(tmp_path, cam, frames)
    common = ["-l", log, "-c", calib, "-n", "96", "-w", str(cam.cols), "-h", str(cam.rows), "-s", "7", "-t", "3"]
    a = _summary(_run(common + ["-o", str(tmp_path / "drv"), "-pcd"], str(tmp_path)))
    got, out = _controller(common + ["-o", str(tmp_path / "ctl"), "-stage"], str(tmp_path))
    assert _announced(out, "TrackerInterfaceThread started", "TrackerInterfaceThread ended", "CloudSliceProcessorThread ended"), out
    assert got["finished"] == "1" and int(got["frames"]) == a["frames"] == len(frames) and int(got["slices"]) == a["slices"] >= 3
    assert int(got["consumed"]) == a["slices"] + 1                       # + the processor's own FIRST slice
    assert open(str(tmp_path / "drv.poses"), "rb").read() == open(str(tmp_path / "ctl.poses"), "rb").read()
    pa, pb = klg.read_pcd(str(tmp_path / "drv.pcd")), klg.read_pcd(str(tmp_path / "ctl.pcd"))
    assert len(pa) == len(pb) == int(got["saved"]) > 1000
    # (extraction order inside a slice is free, so a leaf's float sum may differ in its last bits between two runs: tests/test_pcd.py)
    assert np.abs(pa["xyz"] - pb["xyz"]).max() <= 2e-6 * max(1.0, float(np.abs(pa["xyz"]).max()))
    # the processor thread doing the per-slice stage itself (kt_slice_process on its own context) gives the same cloud
    got2, _ = _controller(common + ["-o", str(tmp_path / "ctl2")], str(tmp_path))
    assert got2["saved"] == got["saved"] and got2["consumed"] == got["consumed"]
    # the End button: endRequested after 8 frames -> finalise, hand-shake, save; fewer frames than the log holds
    got3, _ = _controller(common + ["-o", str(tmp_path / "ctl3"), "-end", "8"], str(tmp_path))
    assert got3["finished"] == "1" and 8 <= int(got3["frames"]) < len(frames) and int(got3["saved"]) > 0
    P = _poses(str(tmp_path / "ctl3.poses"))
    assert len(P) == int(got3["frames"]) - 1 and np.array_equal(P, _poses(str(tmp_path / "ctl.poses"))[:len(P)])   # (the first frame writes no line)
    # the 30 Hz throttle (ThreadDataPack::limit, TrackerInterface.cpp:106-110)
    # -- judged by the child's own steady clock (frames over the time between mainLoop() and join()), not by this process's wall clock:
    # a throttled run cannot exceed 30 frames/s however slow or fast the box is; the unthrottled runs above are far beyond it
    got4, _ = _controller(common + ["-o", str(tmp_path / "ctl4"), "-end", "6", "-limit"], str(tmp_path))
    assert got4["finished"] == "1" and int(got4["frames"]) >= 6 and got4["fps"] <= 30.5
