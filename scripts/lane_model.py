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


# ---- dealing of the tasks to the SIMDs (32x2, 16-z tasks, batches of 4, in-image hull + the exact hull as brackets) ----------------
# kt_tsdf_tasks_kernel's order (wave-columns in index order, the chunks of two x-neighbours interleaved), XCD k takes the k-th eighth,
# local wave = i % 1024 of the XCD's tasks, workgroup j = local wave / 4 on CU j % 32 (dispatch order; a performance assumption only).
# cost of a task in "batch units": set-up 0.7 + 1 per batch that updates a voxel, 0.4 per batch that does not (profiles/r02_tsdf23_whatif.md)
def task_list(z0, z1, m):
    WX, WY, ZC, B = 32, 2, 16, 4
    a0 = z0.reshape(N // WY, WY, N // WX, WX).min((1, 3))
    a1 = z1.reshape(N // WY, WY, N // WX, WX).max((1, 3))
    anyupd = m.reshape(N // B, B, N // WY, WY, N // WX, WX).any((1, 3, 5))      # [z batch][yg][xg]
    XG, M = N // WX, (N // WX) * (N // WY)
    per = ((M + 1023) // 1024 + 1) & ~1
    out = []
    for t in range(1024):
        i0, i1 = min(M, t * per), min(M, t * per + per)
        for i in range(i0, i1, 2):
            rng = []
            for k in (0, 1):
                yg, xg = divmod(i + k, XG)
                lo, hi = int(a0[yg, xg]), int(a1[yg, xg])
                rng.append((yg, xg, lo, hi) if i + k < i1 and lo < hi else None)
            cs = [c for r in rng if r for c in range(r[2] // ZC, (r[3] - 1) // ZC + 1)]
            for c in sorted(set(cs)):
                for r in rng:
                    if r and r[2] // ZC <= c <= (r[3] - 1) // ZC:
                        za, zb = max(r[2], c * ZC), min(r[3], (c + 1) * ZC)
                        nb = -(-(zb - za) // B)
                        hot = sum(bool(anyupd[min((za + q * B) // B, N // B - 1), r[0], r[1]]) for q in range(nb))
                        out.append((0.7 + hot + 0.4 * (nb - hot), nb))
    out = np.array(out)
    return out[:, 0], out[:, 1]


def simd_sums(cost, key, order, split_by_key=False):
    T = len(cost)
    sums = np.zeros((8, 128))
    cum = np.concatenate(([0.0], np.cumsum(key)))
    cuts = [int(np.searchsorted(cum, cum[-1] * x / 8)) for x in range(9)] if split_by_key else [x * T // 8 for x in range(9)]
    for x in range(8):
        idx = order(np.arange(cuts[x], cuts[x + 1]), key)
        for pos, i in enumerate(idx):
            lw = pos % 1024
            sums[x, (lw // 4 % 32) * 4 + lw % 4] += cost[i]
    return sums.ravel()


def by_cost(idx, cost):      # stable: most expensive first, so that round r of the dispatch gets the r-th cost octile
    return idx[np.argsort(-cost[idx], kind="stable")]


def snake(idx, cost):        # sorted, then dealt boustrophedon over the 128 SIMDs of the XCD
    s = idx[np.argsort(-cost[idx], kind="stable")]
    out = np.empty_like(s)
    n = len(s)
    rows = -(-n // 128)
    k = 0
    for r in range(rows):
        seg = s[r * 128:(r + 1) * 128]
        out[k:k + len(seg)] = seg if r % 2 == 0 else seg[::-1]
        k += len(seg)
    # position p of `out` must land on SIMD (p // 4 % 32) * 4 + p % 4: a permutation of 0..127 per row, the same for every row
    return out


for name, m in (("exact hull", upd), ("in-image hull", inimg)):
    z0, z1 = hull(m)
    cost, nb = task_list(z0, z1, upd)
    for oname, o, key, sp in (("shipped order", lambda i, c: i, cost, False), ("sorted by cost", by_cost, cost, False), ("sorted by cost, snake", snake, cost, False),
                              ("sorted by batches", by_cost, nb, False), ("sorted by batches, snake", snake, nb, False),
                              ("XCD split by batches, sorted by batches, snake", snake, nb, True)):
        s = simd_sums(cost, key, o, sp)
        print("dealing, %s, %s: %d tasks, SIMD sums mean %.2f p90 %.2f max %.2f (max / mean %.2f)" % (name, oname, len(cost), s.mean(), np.percentile(s, 90), s.max(), s.max() / s.mean()))
