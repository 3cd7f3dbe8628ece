"""oracle == reference at the bench's own size: 640x480 frames of the orbit sequence integrated into a 512^3 volume (with a storage wrap),
raycast from the next pose, an ICP reduction and the RGB-D residual + step at full resolution, the whole volume extracted, then the kernels of a volume shift on every axis (the 18-plane slab extracted, tsdf and colour slabs cleared forward and back)
-- every output bit for bit.  Minutes of CPU and ~2 GB; not part of the pytest run.
python tests/tools/full_size_pin.py [frames] [orbit512|crabwalk512|farwall768] [quick]      (needs /root/reference: oracle/_ref)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_oracle_vs_ref as T
from oracle import oracle as O, ref as R
from oracle.oracle import OIntr
from kintinuous_amd import synth
R.build(); R.lib(); O.build(); O.lib()
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 3
quick = "quick" in sys.argv[2:]                              # the CPU suite's run: without the shift kernels at the end
big = len(sys.argv) > 2 and sys.argv[2] == "farwall768"     # BASELINE configs[4]: 1280x960 into 768^3 (6 GB)
crab = len(sys.argv) > 2 and sys.argv[2] == "crabwalk512"   # BASELINE configs[2]: the shifting crab-walk, 7 m volume
N, size = (768, 6.0) if big else ((512, 7.0) if crab else (512, 6.0))
cam = synth.Camera.scaled(2 if big else 1)
_, frames, traj, _ = synth.sequence("farwall" if big else ("crabwalk" if crab else "orbit"), nf + 1, cam, 1234)
intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
trunc = max(0.06 if size == 6.0 else max(0.01, size / 100), 2.1 * size / N)
basis = np.array([size / 2, size / 2, -0.45 if big else size / 2], np.float32)   # static mode looks into the volume from 0.45 m in front of it
wrap = [37, N - 11, 130]
vo, co = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
vr, cr = vo.copy(), co.copy()
ok = True
for k in range(nf):
    d, c = frames[k]
    Rm, c0 = traj[k]
    Rk = np.asarray(Rm, np.float32); tk = (np.asarray(c0, np.float32) + basis).astype(np.float32)
    n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
    Rinv = O.mat33_inverse(Rk)
    t0 = time.time()
    U, so = O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, wrap, co, c, n, True)
    t1 = time.time()
    sr = R.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vr, wrap, cr, c, n, True)
    t2 = time.time()
    same = T.same(so, sr) and T.same(vo, vr) and T.same(co, cr)
    ok &= same
    print(f"frame {k}: U {U}  oracle {t1 - t0:.1f} s  reference kernels {t2 - t1:.1f} s  identical {same}", flush=True)
Rm, c0 = traj[nf]
Rk = np.asarray(Rm, np.float32); tk = (np.asarray(c0, np.float32) + basis).astype(np.float32)
outs = []
for M, vol, col in ((O, vo, co), (R, vr, cr)):
    vm, nm = np.full((3 * cam.rows, cam.cols), 7.0, np.float32), np.full((3 * cam.rows, cam.cols), -3.0, np.float32)
    cm = np.full((cam.rows, cam.cols, 4), 9, np.uint8)
    t0 = time.time()
    M.raycast(intr, Rk, tk, trunc, [size] * 3, vol, vm, nm, wrap, cm, col)
    outs.append((vm, nm, cm, time.time() - t0))
(a, b, c_, ta), (a2, b2, c2, tb) = outs
same = T.same(a, a2) and T.same(b, b2) and T.same(c_[..., :3], c2[..., :3])
ok &= same
print(f"raycast: hits {int(np.isfinite(a[:cam.rows]).sum())}  oracle {ta:.1f} s  reference kernel {tb:.1f} s  identical {same}", flush=True)
# one ICP reduction at full resolution against the prediction, from a slightly wrong pose
from conftest import random_rotation
rng = np.random.default_rng(5)
d, c = frames[nf]
vcur = O.create_vmap(intr, O.bilateral_filter(d)); ncur = O.create_nmap(vcur)
Rc = (random_rotation(rng, 0.01) @ Rk).astype(np.float32); tc = (tk + rng.uniform(-0.01, 0.01, 3)).astype(np.float32)
th = float(np.sin(np.float32(20.0 * 3.14159265 / 180.0)))
Ao, bo, ro = O.icp_step(Rc, tc, vcur, ncur, O.mat33_inverse(Rk), tk, intr, a, b, 0.10, th, 0)
Ar, br, rr = R.icp_step(Rc, tc, vcur, ncur, O.mat33_inverse(Rk), tk, intr, a, b, 0.10, th)
same = T.same(Ao, Ar) and T.same(bo, br) and T.same(ro, rr)
ok &= same
print(f"ICP reduction at {cam.cols}x{cam.rows}: inliers {np.asarray(ro).ravel()[1]:.0f}  identical {same}", flush=True)
# the photometric side at full resolution: residual search, sigma / count, Jacobian reduction between the last two frames
(dl, cl), (dn, cn) = frames[nf - 1], frames[nf]
ld, nd_ = O.depth_to_metres(dl, 6000), O.depth_to_metres(dn, 6000)
li, ni = O.bgr_to_intensity(cl), O.bgr_to_intensity(cn)
same = T.same(ld, R.depth_to_metres(dl, 6000)) and T.same(ni, R.bgr_to_intensity(cn))
dx, dy = O.derivative_images(ni)
dxr, dyr = R.derivative_images(ni)
same &= T.same(dx, dxr) and T.same(dy, dyr)
K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], np.float64)
krkinv = (K @ O.rodrigues(rng.uniform(-0.004, 0.004, 3)) @ np.linalg.inv(K)).astype(np.float32)
kt = (K @ rng.uniform(-0.004, 0.004, 3)).astype(np.float32)
min_scale = (np.float32(5) / np.float32(0.125)) ** 2
co_, so_, no_ = O.rgb_residual(min_scale, dx, dy, ld, nd_, li, ni, 0.07, kt, krkinv)
cr_, sr_, nr_ = R.rgb_residual(min_scale, dx, dy, ld, nd_, li, ni, 0.07, kt, krkinv)
msk = co_["valid"] != 0
same &= (so_, no_) == (sr_, nr_) and np.array_equal(msk, cr_["valid"] != 0) and all(T.same(co_[f][msk], cr_[f][msk]) for f in ("zero", "one", "diff"))
if no_:
    cloud = O.project_to_cloud(ld, cam.fx, cam.fy, cam.cx, cam.cy, 0)
    same &= T.same(cloud, R.project_to_cloud(ld, cam.fx, cam.fy, cam.cx, cam.cy, 0))
    Ao, bo = O.rgb_step(co_, float(np.sqrt(np.float32(no_))), cloud, cam.fx, cam.fy, dx, dy, 0.125, 0)
    Ar, br = R.rgb_step(cr_, float(np.sqrt(np.float32(no_))), cloud, cam.fx, cam.fy, dx, dy, 0.125)
    same &= T.same(Ao, Ar) and T.same(bo, br)
ok &= bool(same)
print(f"RGB-D residual + step at {cam.cols}x{cam.rows}: correspondences {no_}  identical {bool(same)}", flush=True)
po = O.extract_cloud_slice(vo, [size] * 3, 12000000, wrap, co, 0, N, 0, N, 0, N, 1, [37, -11, 642])
pr = R.extract_cloud_slice(vr, [size] * 3, 12000000, wrap, cr, 0, N, 0, N, 0, N, 1, [37, -11, 642])
same = len(po) == len(pr) and T._point_set(po) == T._point_set(pr)
ok &= same
print(f"extraction of the whole volume: {len(po)} points  identical {same}")
# a volume shift's kernels on every axis: the 16 + 2 plane slab extracted, then cleared (tsdf and colour volumes), forward and back
for axis in ([] if quick else range(3)):
    best = (-1, 0)
    for start in range(0, N - 18, 36):   # the slab position with the most surface in it (oracle only: cheap)
        lo, hi = [0, 0, 0], [N, N, N]
        lo[axis], hi[axis] = start, start + 18
        best = max(best, (len(O.extract_cloud_slice(vo, [size] * 3, 12000000, wrap, co, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, [37, -11, 642])), start))
    lo, hi = [0, 0, 0], [N, N, N]
    lo[axis], hi[axis] = best[1], best[1] + 18
    po = O.extract_cloud_slice(vo, [size] * 3, 12000000, wrap, co, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, [37, -11, 642])
    pr = R.extract_cloud_slice(vr, [size] * 3, 12000000, wrap, cr, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], 1, [37, -11, 642])
    same = len(po) == len(pr) and T._point_set(po) == T._point_set(pr)
    for back in (False, True):
        cur = wrap[axis] + (N if back else 0)
        delta = cur + (-14 if back else 14)
        for vol_o, vol_r in ((vo, vr), (co.view(np.uint32).reshape(N, N, N), cr.view(np.uint32).reshape(N, N, N))):
            O.clear_volume(vol_o, axis, back, cur, delta)
            R.clear_volume(vol_r, axis, back, cur, delta)
            same &= T.same(vol_o, vol_r)
    ok &= bool(same)
    print(f"shift kernels, axis {axis}: slab of {len(po)} points, 4 clears  identical {bool(same)}", flush=True)
print("PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
