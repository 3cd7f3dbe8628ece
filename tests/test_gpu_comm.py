"""GPU: the path's single collective through the C-ABI (kt_comm_unique_id / kt_comm_init / kt_pose_gather / kt_comm_destroy: one
RCCL all-gather of dense poses).  A gpurun box has one GPU, so this is a one-rank communicator -- the same calls, the same
ncclAllGather; the world-size-2 layout is covered on CPU by tests/test_multigpu_gloo.py and the N-GPU launch is scripts/launch_8gpu.sh."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pose_gather_one_rank(ctx, small_scene):
    from kintinuous_amd import abi
    cam, frames, _ = small_scene
    trk = abi.Tracker(ctx, abi.TrackerConfig(cam.cols, cam.rows, 64, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0))
    for k, (d, rgb) in enumerate(frames[:4]):
        trk.process_frame_host(d, rgb, 33333 * k)
    comm = abi.Comm(ctx, 0, 1)
    comm.barrier()           # kt_comm_barrier: a one-float all-gather on the communicator's stream (bench.py's barriers at N > 1)
    for k in (4, 2):
        got = comm.gather_poses(trk, k)
        want = np.stack([trk.dense_pose(trk.num_poses() - k + i)[1].reshape(16) for i in range(k)])
        assert got.shape == (1, k, 16) and np.array_equal(got[0], want)
    comm.barrier()
    comm.close()
    trk.close()


def test_bench_protocol_one_rank_with_the_communicator(ctx, small_scene):
    """multistream.timed_region (bench.py's protocol) with the real kt_comm on one rank: the gather runs inside the region and the
    rank finds its own poses in it."""
    from kintinuous_amd import abi
    from kintinuous_amd.multistream import check_gather, make_comm, make_exchange, timed_region
    cam, frames, _ = small_scene
    trk = abi.Tracker(ctx, abi.TrackerConfig(cam.cols, cam.rows, 64, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0))
    ex = make_exchange(0, 1)
    comm = make_comm(ex, ctx, 0, 1)
    step = lambda i: trk.process_frame_host(frames[i][0], frames[i][1], 33333 * i)
    region = timed_region(comm, ex, 1, ctx.sync, step, 3, 2, lambda: comm.gather_poses(trk, 3))
    mine = np.stack([trk.dense_pose(trk.num_poses() - 3 + i)[1].reshape(16) for i in range(3)])
    check_gather(region["gathered"], 0, mine)
    assert region["gathered"].shape == (1, 3, 16) and region["fps"] > 0
    comm.close()
    trk.close()
