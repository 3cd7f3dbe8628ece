"""Where do the lane-steps of tsdf23 go?  Runs a bench workload with the counting kernel variant and prints, per frame,
U (voxels updated), voxel steps that project into the image, and lane-steps executed (wave batches x 4 x 64).
usage: python scripts/lane_efficiency.py [workload] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kintinuous_amd import abi, synth

w = sys.argv[1] if len(sys.argv) > 1 else "orbit512"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg_name, scale, N, kw = bench.WORKLOADS[w]
cam = synth.Camera.scaled(scale)
_, frames, traj, kw2 = synth.sequence(cfg_name, n, cam, 1234)
d = dict(volume_size=6.0, voxel_shift=14, overlap=2, static_mode=0, use_rgbd=0, use_rgbd_icp=0, fast_odometry=0, disable_color_angle=0)
d.update(kw2); d.update(kw)
ctx = abi.Ctx(0)
cfg = abi.TrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, d["volume_size"], d["voxel_shift"], d["overlap"], d["static_mode"],
                        d["use_rgbd"], d["use_rgbd_icp"], d["fast_odometry"], d["disable_color_angle"], 0)
trk = abi.Tracker(ctx, cfg)
trk.enable_counts(True)
tot = np.zeros(4)
for i, (dep, rgb) in enumerate(frames):
    trk.process_frame(ctx.upload(dep), ctx.upload(rgb), 33333 * i)
    trk.num_poses()
    c = trk.debug_counts()
    U, batches, tasks, img = c[0], c[1], c[2], c[3]
    lanes = batches * 4 * 64
    if i % 5 == 0 or i < 3:
        print(f"frame {i}: tasks {tasks} lane-steps {lanes/1e6:.2f}M in-image {img/1e6:.2f}M ({img/max(lanes,1):.0%}) updated {U/1e6:.2f}M ({U/max(lanes,1):.0%})")
    if i > 0:
        tot += (lanes, img, U, tasks)
print("mean: lane-steps %.2fM in-image %.0f%% updated %.0f%% tasks %.0f" % (tot[0] / (n - 1) / 1e6, 100 * tot[1] / tot[0], 100 * tot[2] / tot[0], tot[3] / (n - 1)))
