"""Multi-GPU layout of the hot path: independent RGB-D streams, one tracker per GPU (rank), no data-path collective;
the only exchange is ONE gather of per-stream poses through the C-ABI's communicator (kt_comm over RCCL).
bench.py's N-rank protocol lives here (timed_region) so that the CPU test drives the very same control flow with a gloo stand-in for
the communicator (tests/test_multigpu_gloo.py); rank 0's communicator id and the max-over-ranks time travel through a key-value
store, so a job holds exactly one communicator."""
from __future__ import annotations


def stream_seed(rank: int, base: int = 1234) -> int:
    """Deterministic scene seed of the stream a rank owns (stream s -> GPU s)."""
    return base + rank


def pingpong(i: int, n: int) -> int:
    """Frame schedule 0,1,..,n-1,n-2,..,1,0,1,..: a continuous camera motion for any number of steps."""
    if n < 2:
        return 0
    period = 2 * (n - 1)
    j = i % period
    return j if j < n else period - j


class LocalExchange:
    """Out-of-band exchange between ranks, one-rank case."""

    def share(self, key: str, data: bytes) -> bytes:
        return data

    def max(self, key: str, value: float) -> float:
        return value


class StoreExchange:
    """Out-of-band exchange of a few small values between the ranks of one job through a torch.distributed TCPStore -- a key-value
    server, NOT a process group: the job's only communicator is the C-ABI's (kt_comm over RCCL), which needs rank 0's 128-byte id
    delivered before it exists.  Under torch.distributed.run the launcher's own store (MASTER_ADDR:MASTER_PORT) is used as a client;
    launched by hand, rank 0 hosts one."""

    def __init__(self, rank: int, world: int, addr: str = None, port: int = None, timeout_s: float = None):
        import datetime
        import os
        if timeout_s is None:
            timeout_s = float(os.environ.get("KT_EXCHANGE_TIMEOUT_S", "300"))
        from torch.distributed import TCPStore
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("MASTER_PORT", "29533"))
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
        self.rank, self.world = rank, world
        self.store = TCPStore(addr, port, None if agent else world, is_master=(rank == 0 and not agent), wait_for_workers=False,
                              timeout=datetime.timedelta(seconds=timeout_s))
        self.prefix = "kt/%s/" % os.environ.get("TORCHELASTIC_RUN_ID", "job")
        self.generation = {}   # per key: how often it has been exchanged (every rank calls in the same order, so the counts agree)

    def _key(self, key: str) -> str:
        """a fresh store key per exchange of `key`: the store never forgets, and a second timed_region in one job must not read the
        first one's values"""
        g = self.generation.get(key, 0)
        self.generation[key] = g + 1
        return "%s%s#%d" % (self.prefix, key, g)

    def _get(self, k: str, who: str) -> bytes:
        try:
            return bytes(self.store.get(k))   # blocks until the key exists or the store's timeout expires
        except Exception as e:   # noqa: BLE001 -- a rank that died before it published must end the job with a message, not a hang
            raise RuntimeError("multistream: %s never arrived in the key-value store (key %s): a rank of the job is gone? (%s)" % (who, k, e)) from e

    def share(self, key: str, data: bytes) -> bytes:
        """rank 0's bytes on every rank"""
        k = self._key(key)
        if self.rank == 0:
            self.store.set(k, data)
            return data
        return self._get(k, "rank 0's '%s'" % key)

    def max(self, key: str, value: float) -> float:
        """the largest of the ranks' values, on every rank"""
        k = self._key(key)
        self.store.set("%s/%d" % (k, self.rank), repr(float(value)))
        return max(float(self._get("%s/%d" % (k, r), "rank %d's '%s'" % (r, key)).decode()) for r in range(self.world))


def make_exchange(rank: int, world: int):
    return StoreExchange(rank, world) if world > 1 else LocalExchange()


def make_comm(exchange, ctx, rank: int, world: int):
    """The C-ABI communicator of the pose gather (kt_comm_init over RCCL): rank 0's id travels through `exchange`."""
    from . import abi
    return abi.Comm(ctx, rank, world, lambda ident: exchange.share("comm_id", ident))


def timed_region(comm, exchange, world: int, sync, step, steps: int, warmup: int, gather=None, prepare=None):
    """bench.py's measurement protocol for one rank of `world` (one independent stream per rank, weak scaling): `warmup` untimed steps;
    barrier + device sync; EXACTLY `steps` timed steps; the path's single collective (`gather()`, the pose all-gather) inside the
    region; device sync + barrier; the slowest rank's time defines the rate.  `prepare()` runs between the warm-up and the first
    barrier (arming profilers, resetting statistics).  `comm` has abi.Comm's interface (barrier / gather_poses
    / close; None: no collective), `exchange` carries the max over ranks.  Returns {elapsed (max over ranks), local_elapsed, marks,
    gathered, fps}."""
    import time
    for i in range(warmup):
        step(i)
    sync()
    if prepare is not None:
        prepare()
    if comm is not None and world > 1:
        comm.barrier()
    sync()
    t0 = time.perf_counter()
    marks = [t0]
    for i in range(warmup, warmup + steps):
        step(i)
        marks.append(time.perf_counter())
    gathered = gather() if (gather is not None and comm is not None) else None
    sync()
    if comm is not None and world > 1:
        comm.barrier()
    local = time.perf_counter() - t0
    slowest = exchange.max("elapsed", local)
    return {"elapsed": slowest, "local_elapsed": local, "marks": marks, "gathered": gathered, "fps": world * steps / slowest}


def check_gather(gathered, rank: int, mine) -> None:
    """every rank finds its own poses at its own place in the gathered array"""
    import numpy as np
    if not np.array_equal(np.asarray(gathered[rank]), np.asarray(mine)):
        raise AssertionError("the gathered poses are not this rank's")
