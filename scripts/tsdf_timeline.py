"""In-kernel timeline of the voxel kernel (kt_tsdf23_lean_kernel), orbit512, one launch: when waves enter, how long the table fill and a
task's set-up take, how long each batch takes, when SIMDs / XCDs finish.  Needs a library built with -DKT_TSDF_TIMELINE
(scripts/exp_variants.sh build "-DKT_TSDF_TIMELINE"; KT_HIP_LIB=exp/libkt_exp_1.so)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "orbit512"
if wl == "orbit512":
    cam = synth.Camera()
    _, frames, traj, kw = synth.sequence("orbit", 8, cam)
    N = 512
else:
    cam = synth.Camera.scaled(2)
    _, frames, traj, kw = synth.sequence("farwall", 5, cam)
    kw = dict(kw, static_mode=1)
    N = 768
ctx = abi.Ctx(0)
cfg = abi.TrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, kw.get("volume_size", 6.0), 14, 2, 0, 0, 0, 0, kw.get("static_mode", 0), 0)
trk = abi.Tracker(ctx, cfg)
for k, (d, rgb) in enumerate(frames):
    trk.process_frame_host(d, rgb, k)
trk.pose()
W = 8192
buf = (C.c_ulonglong * (W * 16))()
words = abi.lib().kt_debug_tsdf_timeline(ctx.h, buf, W * 16)
assert words > 0, abi.lib().kt_last_error()
T = np.frombuffer(buf, dtype=np.uint64).reshape(W, words).astype(np.int64)
hw = T[:, 0]
st = T[:, 1:]
t0 = st[:, 0][st[:, 0] > 0].min()
us = lambda x: (x - t0) * 0.01
nst = (st > 0).sum(axis=1)                       # stamps per wave: entry, tables, [setup, batches...], exit
has_task = nst >= 5
print("waves", W, "with a task", int(has_task.sum()), "stamps per wave (median)", int(np.median(nst[has_task])))
pct = lambda a: "p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(a, [10, 50, 90, 99, 100]))
print("entry (us after the first wave):", pct(us(st[:, 0])))
print("tables ready - entry:", pct((st[:, 1] - st[:, 0]) * 0.01))
ht = st[has_task]
print("task set up - tables ready:", pct((ht[:, 2] - ht[:, 1]) * 0.01))
for b in range(4):
    ok = (ht[:, 3 + b] > 0) & (ht[:, 4 + b] > 0)   # a following stamp exists: this one is a batch end, not the exit
    if ok.sum():
        print("batch %d:" % (b + 1), int(ok.sum()), "waves,", pct((ht[ok, 3 + b] - ht[ok, 2 + b]) * 0.01))
last = np.array([r[r > 0][-1] for r in st])
print("wave exit:", pct(us(last)))
ent = st[:, 0]
q = np.percentile(ent[has_task], [25, 50, 75])
for lo, hi, name in [(-1, q[0], "first"), (q[0], q[1], "second"), (q[1], q[2], "third"), (q[2], 1 << 62, "last")]:
    sel = has_task & (ent > lo) & (ent <= hi)
    print("waves of the %s entry quartile: entry %.2f-%.2f us, exit" % (name, us(ent[sel].min()), us(ent[sel].max())), pct(us(last[sel])))
print("wave lifetime (with a task):", pct((last[has_task] - st[has_task, 0]) * 0.01), "mean %.2f" % ((last[has_task] - st[has_task, 0]).mean() * 0.01))
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se 3 bits?), ... ; group by the bits that identify a SIMD
simd_key = hw & 0xFFF0 | ((hw >> 16) & 0xF) << 16   # everything but the wave slot (best effort; XCC id lives elsewhere: waves of a workgroup share it)
WPB = int(os.environ.get('KT_TL_WPB', '4'))
xcd = (np.arange(W) // WPB) % 8                     # workgroup -> XCD round robin
key = simd_key.astype(np.int64) * 8 + xcd
fin = {}
for k_, e in zip(key, last):
    fin[k_] = max(fin.get(k_, 0), e)
f = np.array(list(fin.values()))
print("distinct SIMDs seen", len(f), "; SIMD finish:", pct(us(f)))
for x in range(8):
    print("XCD", x, "finish %.2f us" % us(last[xcd == x].max()), end="; ")
print()
print("launch span %.2f us" % us(last.max()))
