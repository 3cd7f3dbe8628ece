"""World-size-2 test of the N > 1 path on CPU (gloo): streams shard one per rank with no data-path collective; the only
exchange is the single pose gather and the max-over-ranks timing that bench.py uses."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    import numpy as np, torch, torch.distributed as dist
    from kintinuous_amd import synth
    from kintinuous_amd.multistream import aggregate_fps, gather_poses, stream_seed
    from oracle.oracle import OTrackerConfig, OracleTracker
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cam = synth.Camera.small(80, 64)
    scene = synth.Scene("room", seed=stream_seed(rank))
    traj = synth.orbit_trajectory(3)
    trk = OracleTracker(OTrackerConfig(cam.cols, cam.rows, 32, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0))
    for k, (R, c) in enumerate(traj):
        d, rgb = synth.render(scene, cam, R, c)
        trk.process_frame(d, rgb, k)
    mine = torch.from_numpy(np.stack([trk.dense_pose(i)[1].reshape(16) for i in range(trk.num_poses())]).astype(np.float32))
    allp = gather_poses(dist, mine, world)
    assert allp.shape == (world, 3, 16)
    assert torch.equal(allp[rank], mine)
    other = allp[1 - rank]
    assert torch.allclose(other[0], mine[0])            # every stream starts at the same initial pose
    assert not torch.equal(other[2], mine[2])           # different scenes (seeds) -> different tracked poses
    fps = aggregate_fps(dist, steps=10, elapsed_s=1.0 + rank, world=world)  # slowest rank (2.0 s) defines the rate
    assert abs(fps - world * 10 / 2.0) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(%r, "rank%%d.ok" %% rank), "w").write("ok")
""")


def test_two_rank_pose_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
