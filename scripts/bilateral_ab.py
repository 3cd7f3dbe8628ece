"""A/B of the bilateral kernels (KT_BILATERAL_V1=1: round 1-4's one-pixel-per-thread kernel; default: kt_bilateral2_kernel), alone on the GPU:
back-to-back launches on one stream, wall clock over 200 of them after a warm-up, at 640x480 and 1280x960; and bit-equality of the outputs.
    python scripts/bilateral_ab.py            (spawns itself once per kernel)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import hashlib
    import numpy as np
    from kintinuous_amd import abi, synth
    ctx = abi.Ctx(0)
    out = {}
    for scale in (1, 2):
        cam = synth.Camera.scaled(scale)
        _, frames, _, _ = synth.sequence("orbit", 2, cam)
        d = np.ascontiguousarray(frames[1][0], np.uint16)
        src, dst = ctx.upload(d), ctx.zeros(d.nbytes)
        for _ in range(20):
            ctx.bilateral_filter(src, dst, cam.cols, cam.rows)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.bilateral_filter(src, dst, cam.cols, cam.rows)
        ctx.sync()
        us = 1e6 * (time.perf_counter() - t0) / 200
        out["%dx%d" % (cam.cols, cam.rows)] = (round(us, 2), hashlib.sha256(ctx.download(dst, np.uint16, d.shape).tobytes()).hexdigest()[:16])
    print(os.environ.get("KT_BILATERAL_V1", "0"), out, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for v in ("1", "0", "1", "0"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, KT_BILATERAL_V1=v))
