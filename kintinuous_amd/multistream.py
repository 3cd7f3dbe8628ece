"""Multi-GPU layout of the hot path: independent RGB-D streams, one tracker per GPU (rank), no data-path collective;
the only exchange is ONE gather of per-stream poses.  The helpers here are backend-agnostic
(torch.distributed with nccl == RCCL on the GPU box, gloo in the CPU tests)."""
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


def gather_poses(dist, local_poses, world: int):
    """All ranks contribute a [k, 16] float32 tensor of row-major 4x4 dense poses ([R | currentGlobalCamera],
    KintinuousTracker.h:151-169); returns the [world, k, 16] tensor on every rank (one collective)."""
    import torch
    assert local_poses.dtype == torch.float32 and local_poses.dim() == 2 and local_poses.shape[1] == 16
    out = torch.empty((world,) + tuple(local_poses.shape), dtype=torch.float32, device=local_poses.device)
    dist.all_gather_into_tensor(out.view(-1), local_poses.contiguous().view(-1))
    return out


def make_comm(dist, ctx, rank: int, world: int):
    """The C-ABI communicator of the pose gather (kt_comm_init over RCCL): rank 0's id travels through torch.distributed's object
    broadcast when there is more than one rank."""
    from . import abi

    def share(ident: bytes) -> bytes:
        if dist is None or world <= 1:
            return ident
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    return abi.Comm(ctx, rank, world, share)


def aggregate_fps(dist, steps: int, elapsed_s: float, world: int, device=None) -> float:
    """Whole-job frames/s: all ranks' frames over the slowest rank's time (max over ranks)."""
    if dist is None or world <= 1:
        return world * steps / elapsed_s
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return world * steps / float(t.item())
