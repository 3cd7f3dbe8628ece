"""One of two PROCESSES sharing GPU 0 (tests/test_gpu_two_process.py): the library's defaults, a 320x240 orbit into a 256^3 volume (a sparse
view by the tracker's rule, like 640x480 into 512^3: the odometry takes the level form),
`passes` passes of `frames` frames (reset in between).  Prints one JSON line: the poses of every pass as hex words, a hash of the volumes the last pass left, the number of
odometry fallbacks, and whether the last frame ran the level form.  `--barrier DIR --me K --peers N` makes the workers start their frames
together (files in DIR)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--barrier", default=None)
    ap.add_argument("--me", type=int, default=0)
    ap.add_argument("--peers", type=int, default=1)
    a = ap.parse_args()
    from kintinuous_amd import abi, synth
    cam = synth.Camera.small(320, 240)
    scene = synth.Scene("room")
    frames = [synth.render(scene, cam, R, c) for (R, c) in synth.orbit_trajectory(a.frames)]
    ctx = abi.Ctx(0)
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 256, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
    trk = abi.Tracker(ctx, cfg)
    trk.process_frame_host(frames[0][0], frames[0][1], 0)   # every kernel has had its first launch
    trk.pose()
    trk.reset()
    if a.barrier:
        open(os.path.join(a.barrier, f"ready_{a.me}"), "w").close()
        t0 = time.time()
        while not all(os.path.exists(os.path.join(a.barrier, f"ready_{k}")) for k in range(a.peers)):
            if time.time() - t0 > 120:
                raise SystemExit("two_process_worker: the peer never arrived")
            time.sleep(0.001)
    out = []
    t0 = time.perf_counter()
    for p in range(a.passes):
        if p:
            trk.reset()
        poses = []
        for k, (d, rgb) in enumerate(frames):
            trk.process_frame_host(d, rgb, 33333 * k)
            poses.append(np.concatenate([x.ravel() for x in trk.pose()]).astype(np.float32))
        out.append(np.array(poses).view(np.uint32).ravel().tolist())
    dt = time.perf_counter() - t0
    import hashlib
    vol_sha = hashlib.sha256(trk.volume().tobytes() + trk.color_volume().tobytes()).hexdigest()   # the volumes the last pass left
    res = {"poses": out, "volumes_sha256": vol_sha, "fallbacks": trk.odometry_fallbacks(), "level_form_last": int(abi.lib().kt_tracker_debug_icp_levels(trk.h)), "seconds": dt}
    trk.close()
    ctx.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
