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
    for k in (4, 2):
        got = comm.gather_poses(trk, k)
        want = np.stack([trk.dense_pose(trk.num_poses() - k + i)[1].reshape(16) for i in range(k)])
        assert got.shape == (1, k, 16) and np.array_equal(got[0], want)
    comm.close()
    trk.close()
