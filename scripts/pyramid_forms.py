"""kt_build_pyramid alone at two image sizes: 300 back-to-back launches, wall clock around them (a launch is 10-60 us: the queue never runs dry).
Round 6 ran it per form of the pyramid (a switch KT_PYR_FORM that has gone with the one-launch kernel: 1 = one launch, 17.2 / 28.2 us at 640x480 /
1280x960, with the small tile 17.2 / 43.8; 2 = two launches, 14.7 / 26.2 -- the tree's form; 3 = two launches, small tile in the second: 15.1 / 36.4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi

ctx = abi.Ctx(0)
rng = np.random.default_rng(1)
for cols, rows in ((640, 480), (1280, 960)):
    depth = (1500 + 400 * np.sin(np.arange(cols)[None, :] / 37.0) + 300 * np.cos(np.arange(rows)[:, None] / 23.0) + rng.integers(0, 8, (rows, cols))).astype(np.uint16)
    d0 = ctx.upload(depth)
    dl = [ctx.empty((cols >> l) * (rows >> l) * 2) for l in (1, 2, 3)]
    vm = [ctx.empty((cols >> l) * (rows >> l) * 12) for l in range(4)]
    nm = [ctx.empty((cols >> l) * (rows >> l) * 12) for l in range(4)]
    intr = abi.Intr(525.0 * cols / 640, 525.0 * cols / 640, cols / 2 - 0.5, rows / 2 - 0.5)
    for _ in range(20): ctx.build_pyramid(intr, d0, cols, rows, dl, vm, nm)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(300): ctx.build_pyramid(intr, d0, cols, rows, dl, vm, nm)
    ctx.sync()
    print("%dx%d: %.1f us per pyramid" % (cols, rows, (time.perf_counter() - t0) / 300 * 1e6))
