"""CPU model of where tsdf23's lane-steps go (no GPU needed): evaluates the reference's in-image test and update predicate for every voxel
of one orbit512 frame with numpy, then counts the lane-steps (wave z-steps x 64) of wave-column shapes / z chunks / batch lengths under two
interval models -- the exact hull of the updated voxels of every column (the best any pre-pass could do) and the in-image hull (no depth
prune at all).  The shipped kernel (32x2 wave-columns, 16-z tasks, batches of 4, piecewise depth-range prune) lies between the two:
scripts/lane_efficiency.py measures it on the GPU (profiles/r02_lane_eff_orbit512.log).
usage: python scripts/lane_model.py [frame]      (about a minute, 300 MB)"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import synth

N, size = 512, 6.0
cell = size / N
trunc = max(0.06, 2.1 * cell)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cam = synth.Camera.scaled(1)
_, frames, traj, _ = synth.sequence("orbit", k + 1, cam, 1234)
dep = frames[k][0].astype(np.float32) / 1000.0
R, c = traj[k]
Ri, tc = np.asarray(R, np.float64).T, np.asarray(c) + size / 2
ys, xs = np.mgrid[0:cam.rows, 0:cam.cols]
Dp = (dep * np.sqrt(((xs - cam.cx) / cam.fx) ** 2 + ((ys - cam.cy) / cam.fy) ** 2 + 1)).astype(np.float32)
ax = (np.arange(N) + 0.5) * cell
gx, gy = ax[None, :] - tc[0], ax[:, None] - tc[1]
inimg, upd = np.zeros((N, N, N), bool), np.zeros((N, N, N), bool)
for z in range(N):
    gz = ax[z] - tc[2]
    vx, vy, vz = (Ri[r, 0] * gx + Ri[r, 1] * gy + Ri[r, 2] * gz for r in range(3))
    with np.errstate(all="ignore"):
        u, v = np.rint(vx * cam.fx / vz + cam.cx), np.rint(vy * cam.fy / vz + cam.cy)
    ok = (vz > 0) & (u >= 0) & (u < cam.cols) & (v >= 0) & (v < cam.rows)
    d = Dp[np.where(ok, v, 0).astype(int), np.where(ok, u, 0).astype(int)]
    inimg[z] = ok
    upd[z] = ok & (d != 0) & (d - np.sqrt(gx ** 2 + gy ** 2 + gz ** 2) >= -trunc)
U = int(upd.sum())
print("frame %d: updated voxels U = %.2fM, voxels that project into the image %.2fM" % (k, U / 1e6, inimg.sum() / 1e6))


def hull(m):
    any_ = m.any(0)
    return np.where(any_, m.argmax(0), N), np.where(any_, N - m[::-1].argmax(0), 0)


def lane_steps(z0, z1, WX, WY, ZC, B):
    a0 = z0.reshape(N // WY, WY, N // WX, WX).min((1, 3))
    a1 = z1.reshape(N // WY, WY, N // WX, WX).max((1, 3))
    tot = tasks = 0
    for ch in range(N // ZC):
        ln = np.maximum(np.minimum(a1, (ch + 1) * ZC) - np.maximum(a0, ch * ZC), 0)
        tot += int((np.ceil(ln / B) * B).sum()) * 64
        tasks += int((ln > 0).sum())
    return tot, tasks


for name, m in (("exact hull of the updated voxels per column", upd), ("in-image hull per column (no depth prune)", inimg)):
    z0, z1 = hull(m)
    print("%s: %.2fM column voxels" % (name, np.maximum(z1 - z0, 0).sum() / 1e6))
    for (WX, WY), ZC, B in itertools.product(((32, 2), (64, 1), (16, 4), (8, 8)), (8, 16), (1, 4)):
        t, n = lane_steps(z0, z1, WX, WY, ZC, B)
        print("  %2dx%d columns, %2d-z tasks, batches of %d: %.2fM lane-steps (U / lane-steps %.0f%%), %d tasks" % (WX, WY, ZC, B, t / 1e6, 100.0 * U / t, n))
