"""How far does the HIP integrate kernel follow the oracle on volume states the reference's own kernels cannot produce (weights above 128,
coloured voxels of weight 0, raw -32768)?  Prints mismatch counts by cause; informational (tests/test_gpu_sweep.py covers reachable states)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import random_rotation, random_volume_state
from hip_kernels import HipKernels
from kintinuous_amd import abi, synth
from oracle import oracle as O
from oracle.oracle import OIntr
O.build(); O.lib()
H = HipKernels(abi.Ctx(0))
rng = np.random.default_rng(4242)
N, size, cols, rows = 72, 6.0, 160, 120
cam = synth.Camera.small(cols, rows)
scene = synth.Scene("room", seed=5)
intr = OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
trunc = max(0.06, 2.1 * size / N)
vo, co = random_volume_state(rng, N, reachable=False)
v0, c0_ = vo.copy(), co.copy()
vh, ch = vo.copy(), co.copy()
Rm, c0 = synth.orbit_trajectory(40)[3]
d, c = synth.render(scene, cam, Rm, c0, noise_mm=1.5, rng=rng)
c = rng.integers(0, 256, c.shape).astype(np.uint8)
Rk = np.asarray(Rm, np.float32); tk = (np.asarray(c0, np.float32) + np.float32(size / 2)).astype(np.float32)
n = O.create_nmap(O.create_vmap(intr, O.bilateral_filter(d)))
Rinv = O.mat33_inverse(Rk)
U, so = O.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vo, [0, 0, 0], co, c, n, True)
H.integrate_tsdf(d, intr, [size] * 3, Rinv, tk, trunc, vh, [0, 0, 0], ch, c, n, True)
bv = vo != vh; bc = (co != ch).any(axis=-1)
w0 = c0_[..., 3]
print("updated", U, "tsdf mismatches", int(bv.sum()), "colour-word mismatches", int(bc.sum()))
print("  colour mismatches with stored weight > 128:", int((bc & (w0 > 128)).sum()), " weight 0 and colour != 0:", int((bc & (w0 == 0)).sum()),
      " neither:", int((bc & (w0 <= 128) & (w0 > 0)).sum()))
print("  tsdf mismatches at raw -32768:", int((bv & (v0 == -32768)).sum()), " elsewhere:", int((bv & (v0 != -32768)).sum()))
