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
    slow = [1]                                                    # the rank whose time must define the rate
    die = int(os.environ.get("KT_TEST_DIE_RANK", "-1"))

    def step(i):
        d, rgb = frames[pingpong(i, len(frames))]
        trk.process_frame(d, rgb, 33333 * i)
        if i >= warmup and rank == slow[0]:
            time.sleep(0.05)
        if rank == die and i == warmup + 1:
            os._exit(3)                                           # a rank that dies inside the timed region, before the gather

    region = timed_region(comm, exchange, world, lambda: None, step, steps, warmup, lambda: comm.gather_poses(trk, steps), lambda: prepared.append(1))
    assert prepared == [1] and comm.barriers == 2                # one barrier on each side of the timed region
    allp = region["gathered"]
    assert allp.shape == (world, steps, 16) and len(region["marks"]) == steps + 1
    mine = np.stack([trk.dense_pose(trk.num_poses() - steps + i)[1].reshape(16) for i in range(steps)])
    check_gather(allp, rank, mine)
    other = (rank + 1) %% world
    assert not np.array_equal(allp[other][-1], mine[-1])         # different scenes (seeds) -> different tracked poses
    # max over ranks: every rank reports the slow rank's time
    assert region["elapsed"] >= region["local_elapsed"] - 1e-9 and region["elapsed"] >= 0.15
    slowest = exchange.max("check", region["local_elapsed"])
    assert abs(slowest - region["elapsed"]) < 1e-9
    assert abs(region["fps"] - world * steps / region["elapsed"]) < 1e-9
    try:
        check_gather(allp, other, mine)
        raise SystemExit("check_gather accepted another rank's poses")
    except AssertionError:
        pass
    # a SECOND timed region in the same job: the exchange must not hand back the first region's values (keys carry a generation).
    # Now rank 0 is the slow one, and slower than rank 1 was.
    slow[0] = 0
    def step2(i):
        d, rgb = frames[pingpong(i, len(frames))]
        trk.process_frame(d, rgb, 33333 * (i + 10))
        if i >= warmup and rank == 0:
            time.sleep(0.12)
    region2 = timed_region(comm, exchange, world, lambda: None, step2, steps, warmup, lambda: comm.gather_poses(trk, steps), None)
    assert region2["elapsed"] >= 0.36 and region2["elapsed"] >= region2["local_elapsed"] - 1e-9, (region2["elapsed"], region["elapsed"])
    assert exchange.generation["elapsed"] == 2 and exchange.generation["built"] == 1
    comm.close()
    trk.close()
    open(os.path.join(%r, "rank%%d.ok" %% rank), "w").write("ok")
""")


def _launch(tmp_path, world, port, extra_env=None, timeout=600):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                           "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=timeout, env=env)


def test_two_rank_bench_protocol(tmp_path):
    r = _launch(tmp_path, 2, 29517)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_four_rank_bench_protocol(tmp_path):
    r = _launch(tmp_path, 4, 29518)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert all((tmp_path / ("rank%d.ok" % k)).exists() for k in range(4))


def test_a_rank_that_dies_ends_the_job(tmp_path):
    """Rank 2 of 4 dies inside the timed region, before the gather.  The others sit in the collective (or in the key-value store waiting
    for its time): the job must END with an error within a bounded time -- the launcher takes the survivors down -- not hang, and no
    rank may report success."""
    import time
    t0 = time.time()
    r = _launch(tmp_path, 4, 29519, extra_env={"KT_TEST_DIE_RANK": "2", "KT_EXCHANGE_TIMEOUT_S": "30"}, timeout=300)
    assert r.returncode != 0
    assert time.time() - t0 < 240
    assert not any((tmp_path / ("rank%d.ok" % k)).exists() for k in range(4))


def test_store_exchange_keys_carry_a_generation():
    """two exchanges of the same key in one job use different store keys (a stale value can not be read back); a missing rank raises"""
    import threading
    from kintinuous_amd.multistream import StoreExchange
    port = 29520
    out = {}

    def rank(r):
        ex = StoreExchange(r, 2, "127.0.0.1", port, timeout_s=20)
        a = ex.max("elapsed", 1.0 + r)
        b = ex.max("elapsed", 0.25 - 0.125 * r)       # smaller than anything of the first round
        c = ex.share("id", b"x" if r == 0 else b"")
        out[r] = (a, b, c, dict(ex.generation), ex)   # (rank 0 hosts the store: it must outlive rank 1's last read)

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert out[0][:3] == (2.0, 0.25, b"x") and out[1][:3] == (2.0, 0.25, b"x")
    assert out[0][3] == {"elapsed": 2, "id": 1}


def test_one_rank_exchange_is_local():
    from kintinuous_amd.multistream import LocalExchange, make_exchange, timed_region
    ex = make_exchange(0, 1)
    assert isinstance(ex, LocalExchange) and ex.share("k", b"abc") == b"abc" and ex.max("t", 2.5) == 2.5
    calls = []
    region = timed_region(None, ex, 1, lambda: calls.append("sync"), lambda i: calls.append(i), 3, 2)
    assert calls[:3] == [0, 1, "sync"] and [c for c in calls if c != "sync"] == [0, 1, 2, 3, 4] and region["gathered"] is None
    assert abs(region["fps"] - 3 / region["elapsed"]) < 1e-9


def test_store_exchange_reports_a_missing_rank():
    """a rank that never publishes its value makes the others raise after the store's timeout (bench.py then exits non-zero)"""
    import pytest
    from kintinuous_amd.multistream import StoreExchange
    ex = StoreExchange(0, 2, "127.0.0.1", 29521, timeout_s=2)
    with pytest.raises(RuntimeError, match="rank 1's 'elapsed' never arrived"):
        ex.max("elapsed", 1.0)
