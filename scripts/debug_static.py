import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kintinuous_amd import abi, synth
from oracle import oracle

cam = synth.Camera.small(160, 120)
scene = synth.Scene("farwall")
traj = synth.static_trajectory(4)
frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
ctx = abi.Ctx(0)
g = abi.TrackerConfig(cam.cols, cam.rows, 64, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 1, 0, 0, 0, 0, 0)
o = oracle.OTrackerConfig(cam.cols, cam.rows, 64, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 1, 0, 0, 0, 0, 0)
trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
for k in range(4):
    d, rgb = frames[k]
    trk.process_frame_host(d, rgb, k)
    otr.process_frame(d, rgb, k)
    R, t, _ = trk.pose(); Ro, to, _ = otr.pose()
    print(k, "dt", np.abs(t - to).max(), "dR", np.abs(R - Ro).max())
    v, ov = trk.volume(), otr.volume()
    print("   vol mismatch", int((v != ov).sum()), "colw", int((trk.color_volume() != otr.color_volume()).sum()))
    for lvl in range(4):
        a, b = trk.vmap_g_prev(lvl), otr.vmap_g_prev(lvl)
        rows = a.shape[0] // 3
        va, vb = np.isfinite(a[:rows]), np.isfinite(b[:rows])
        m = va & vb
        print("   lvl", lvl, "valid hip/oracle", int(va.sum()), int(vb.sum()), "mask diff", int((va != vb).sum()),
              "x diff", int((a[:rows][m].view(np.uint32) != b[:rows][m].view(np.uint32)).sum()))
    if k >= 1:
        bad = np.argwhere(np.isfinite(trk.vmap_g_prev(0)[:cam.rows]) != np.isfinite(otr.vmap_g_prev(0)[:cam.rows]))
        print("   first mask diffs", bad[:5].tolist())
