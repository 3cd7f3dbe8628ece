"""CPU model of tsdf23's lane-steps, any bench workload (round 6; scripts/lane_model.py is the orbit512-only original): evaluates the reference's
in-image test and update predicate for every voxel of one frame with numpy, then counts lane-steps (wave z-steps x 64) for the wave-column
shapes under three interval models -- (a) the exact hull of the updated voxels of every column, (b) the hull of "in image and not farther
than the local depth + trunc in a dilated tile" (what a tile-maximum prune can know), (c) the in-image hull -- and, per 32 x 8 super-column,
the best of the three shapes (a mixed tiling).  usage: python scripts/lane_model2.py [orbit512|farwall768] [frame]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import synth

w = sys.argv[1] if len(sys.argv) > 1 else "orbit512"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, scale, cfgname, static = (512, 1, "orbit", False) if w == "orbit512" else (768, 2, "farwall", True)
size = 6.0
cell = size / N
trunc = max(0.06, 2.1 * cell)
cam = synth.Camera.scaled(scale)
_, frames, traj, _ = synth.sequence(cfgname, k + 1, cam, 1234)
dep = frames[k][0].astype(np.float32) / 1000.0
R, c = traj[k]
basis = np.array([size / 2, size / 2, size / 2 - (size / 2 + 0.45) if static else size / 2])
Ri, tc = np.asarray(R, np.float64).T, np.asarray(c) + basis
ys, xs = np.mgrid[0:cam.rows, 0:cam.cols]
Dp = (dep * np.sqrt(((xs - cam.cx) / cam.fx) ** 2 + ((ys - cam.cy) / cam.fy) ** 2 + 1)).astype(np.float32)
# tile maxima of the scaled depth, dilated 3 x 3 (tile = 8 px at VGA, 16 at 1280x960): what the interval pre-pass can look up
T = 8 * scale
tm = Dp.reshape(cam.rows // T, T, cam.cols // T, T).max((1, 3))
pad = np.pad(tm, 1, mode="edge")
tmd = np.max([pad[i:i + tm.shape[0], j:j + tm.shape[1]] for i in range(3) for j in range(3)], axis=0)
ax = (np.arange(N) + 0.5) * cell
gx, gy = ax[None, :] - tc[0], ax[:, None] - tc[1]
z0 = {m: np.full((N, N), N, np.int32) for m in "abc"}
z1 = {m: np.zeros((N, N), np.int32) for m in "abc"}
U = 0
updz = []
for z in range(N):
    gz = ax[z] - tc[2]
    vx, vy, vz = (Ri[r, 0] * gx + Ri[r, 1] * gy + Ri[r, 2] * gz for r in range(3))
    with np.errstate(all="ignore"):
        u, v = np.rint(vx * cam.fx / vz + cam.cx), np.rint(vy * cam.fy / vz + cam.cy)
    ok = (vz > 0) & (u >= 0) & (u < cam.cols) & (v >= 0) & (v < cam.rows)
    ui, vi = np.where(ok, u, 0).astype(int), np.where(ok, v, 0).astype(int)
    d = Dp[vi, ui]
    rng = np.sqrt(gx ** 2 + gy ** 2 + gz ** 2)
    upd = ok & (d != 0) & (d - rng >= -trunc)
    til = ok & (tmd[vi // T, ui // T] - rng >= -trunc)
    U += int(upd.sum())
    for m, mask in (("a", upd), ("b", til), ("c", ok)):
        z0[m] = np.where(mask & (z0[m] == N), z, z0[m])
        z1[m] = np.where(mask, z + 1, z1[m])
    updz.append(np.packbits(upd, axis=1))
print("%s frame %d: U = %.2fM" % (w, k, U / 1e6))


def steps(a0, a1, ZC=16, B=4):
    tot = np.zeros(a0.shape, np.int64)
    for ch in range(N // ZC):
        ln = np.maximum(np.minimum(a1, (ch + 1) * ZC) - np.maximum(a0, ch * ZC), 0)
        tot += (-(-ln // B)) * B * 64
    return tot


names = {"a": "exact hull of the updated voxels", "b": "dilated tile-maximum depth bound", "c": "in-image hull"}
for m in "abc":
    print("%s: %.2fM column voxels" % (names[m], np.maximum(z1[m] - z0[m], 0).sum() / 1e6))
    per = {}
    for WX, WY in ((32, 2), (16, 4), (8, 8)):
        a0 = z0[m].reshape(N // WY, WY, N // WX, WX).min((1, 3))
        a1 = z1[m].reshape(N // WY, WY, N // WX, WX).max((1, 3))
        st = steps(a0, a1)
        per[(WX, WY)] = st.reshape(N // 8, 8 // WY, N // 32, 32 // WX).sum((1, 3))   # per 32 x 8 super-column
        print("  %2dx%d wave-columns: %.2fM lane-steps, U / lane-steps %.1f%%" % (WX, WY, st.sum() / 1e6, 100.0 * U / st.sum()))
    best = np.minimum(np.minimum(per[(32, 2)], per[(16, 4)]), per[(8, 8)])
    print("  best shape per 32x8 super-column: %.2fM lane-steps, U / lane-steps %.1f%%" % (best.sum() / 1e6, 100.0 * U / best.sum()))
