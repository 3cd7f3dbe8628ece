"""World-size-2 test of the N > 1 path on CPU: the ranks run bench.py's OWN measurement protocol (kintinuous_amd.multistream.timed_region:
warm-up, barrier, timed steps, the single pose gather inside the region, barrier, max-over-ranks time through the key-value store,
the gather-then-check of the rank's own poses) with the communicator id shared through the same StoreExchange bench.py uses.  The one
thing that is a stand-in is the communicator: a test-side class with abi.Comm's interface (gather_poses / barrier / close) over a gloo
process group, and the oracle's tracker as the stream (no GPU here).  On the GPU box the communicator is kt_comm over RCCL."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    import numpy as np, torch, torch.distributed as dist
    from kintinuous_amd import synth
    from kintinuous_amd.multistream import check_gather, make_exchange, pingpong, stream_seed, timed_region
    from oracle.oracle import OTrackerConfig, OracleTracker
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

    class GlooComm:
        '''abi.Comm's interface over a gloo group (test side only; the product has no such fallback)'''
        def __init__(self, exchange, rank, world):
            # like abi.Comm: rank 0 makes an id, every rank receives it out of band before the communicator exists
            ident = exchange.share("comm_id", bytes(range(128)) if rank == 0 else b"")
            assert ident == bytes(range(128))
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            self.world = world
            self.barriers = 0
        def gather_poses(self, trk, k):
            mine = torch.from_numpy(np.stack([trk.dense_pose(trk.num_poses() - k + i)[1].reshape(16) for i in range(k)]).astype(np.float32))
            out = torch.empty((self.world, k, 16), dtype=torch.float32)
            dist.all_gather_into_tensor(out.view(-1), mine.contiguous().view(-1))
            return out.numpy()
        def barrier(self):
            self.barriers += 1
            dist.barrier()
        def close(self):
            dist.destroy_process_group()

    exchange = make_exchange(rank, world)
    assert exchange.share("built", b"1") == b"1"
    cam = synth.Camera.small(80, 64)
    scene = synth.Scene("room", seed=stream_seed(rank))          # one independent stream per rank
    traj = synth.orbit_trajectory(4)
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    trk = OracleTracker(OTrackerConfig(cam.cols, cam.rows, 32, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0))
    comm = GlooComm(exchange, rank, world)
    steps, warmup, prepared = 3, 1, []

    def step(i):
        d, rgb = frames[pingpong(i, len(frames))]
        trk.process_frame(d, rgb, 33333 * i)
        if i >= warmup and rank == 1:
            time.sleep(0.05)                                      # rank 1 is the slow one: its time must define the rate

    region = timed_region(comm, exchange, world, lambda: None, step, steps, warmup, lambda: comm.gather_poses(trk, steps), lambda: prepared.append(1))
    assert prepared == [1] and comm.barriers == 2                # one barrier on each side of the timed region
    allp = region["gathered"]
    assert allp.shape == (world, steps, 16) and len(region["marks"]) == steps + 1
    mine = np.stack([trk.dense_pose(trk.num_poses() - steps + i)[1].reshape(16) for i in range(steps)])
    check_gather(allp, rank, mine)
    assert not np.array_equal(allp[1 - rank][-1], mine[-1])      # different scenes (seeds) -> different tracked poses
    # max over ranks: both ranks report the slow rank's time
    assert region["elapsed"] >= region["local_elapsed"] - 1e-9 and region["elapsed"] >= 0.15
    slowest = exchange.max("check", region["local_elapsed"])
    assert abs(slowest - region["elapsed"]) < 1e-9
    assert abs(region["fps"] - world * steps / region["elapsed"]) < 1e-9
    try:
        check_gather(allp, 1 - rank, mine)
        raise SystemExit("check_gather accepted another rank's poses")
    except AssertionError:
        pass
    comm.close()
    trk.close()
    open(os.path.join(%r, "rank%%d.ok" %% rank), "w").write("ok")
""")


def test_two_rank_bench_protocol(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_one_rank_exchange_is_local():
    from kintinuous_amd.multistream import LocalExchange, make_exchange, timed_region
    ex = make_exchange(0, 1)
    assert isinstance(ex, LocalExchange) and ex.share("k", b"abc") == b"abc" and ex.max("t", 2.5) == 2.5
    calls = []
    region = timed_region(None, ex, 1, lambda: calls.append("sync"), lambda i: calls.append(i), 3, 2)
    assert calls[:3] == [0, 1, "sync"] and [c for c in calls if c != "sync"] == [0, 1, 2, 3, 4] and region["gathered"] is None
    assert abs(region["fps"] - 3 / region["elapsed"]) < 1e-9
