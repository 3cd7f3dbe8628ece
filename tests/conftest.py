import os
import sys

# the oracle is OpenMP code: cap its threads (GPU boxes expose many more hardware threads than the job may use)
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, len(os.sched_getaffinity(0)))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of a run (round 5).  The driver runs the GPU suite with -x: a failure hides everything behind it, so the kernel-parity evidence goes
# FIRST (reference-made goldens, the BASELINE configs, then per-kernel modules) and the process-level tests -- child processes, threads,
# files: the only tests whose outcome can depend on scheduling -- go LAST.  Unknown modules keep their place in the middle.
_ORDER = ["test_golden_ref", "test_golden", "test_gpu_configs", "test_gpu_fullsize", "test_gpu_image", "test_gpu_track", "test_gpu_solve",
          "test_gpu_volume", "test_gpu_sweep", "test_gpu_tol", "test_gpu_tracker", "test_slice_process", "test_pcd", "test_jpeg", "test_gpu_comm"]
_LAST = ["test_gpu_bench_cli", "test_gpu_two_process", "test_gpu_host_shell"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = item.module.__name__.split(".")[-1]
        if name in _LAST:
            return 1000 + _LAST.index(name)
        return _ORDER.index(name) if name in _ORDER else len(_ORDER)
    items.sort(key=rank)          # stable: the order inside a module is kept


# Round 4: two voxel kernels store the same bits (kt_tsdf23_lean_kernel and the round-3 kt_tsdf23_kernel; kt_debug_tsdf_lean selects):
# every GPU test of these modules runs once per kernel.
VOXEL_KERNEL_MODULES = {"test_gpu_volume", "test_gpu_sweep", "test_golden_ref", "test_golden", "test_gpu_fullsize"}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in VOXEL_KERNEL_MODULES and metafunc.definition.get_closest_marker("gpu"):
        if "voxel_kernel" not in metafunc.fixturenames:
            metafunc.fixturenames.append("voxel_kernel")
        metafunc.parametrize("voxel_kernel", ["lean", "r3"], indirect=True)


@pytest.fixture
def voxel_kernel(request, ktlib):
    from kintinuous_amd import abi
    abi._chk(ktlib.kt_debug_tsdf_lean(1 if request.param == "lean" else 0))
    yield request.param
    abi._chk(ktlib.kt_debug_tsdf_lean(-1))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ktlib():
    """The HIP library.  Fails loudly (no fallback) when it is missing."""
    from kintinuous_amd import abi
    return abi.lib()


@pytest.fixture(scope="session")
def ctx(ktlib):
    from kintinuous_amd import abi
    c = abi.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def small_scene():
    """A 160x120 room frame pair + maps, shared by several tests."""
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("room")
    traj = synth.orbit_trajectory(8)
    frames = [synth.render(scene, cam, R, c) for (R, c) in traj]
    return cam, frames, traj


def random_rotation(rng, max_angle):
    from oracle import oracle
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    return oracle.rodrigues(axis * rng.uniform(0, max_angle)).astype(np.float32)


def random_volume_state(rng, N, reachable):
    """A volume in a random STATE instead of one grown from a cleared volume: every tsdf word, colour and weight drawn at random, so that
    one integrate call puts a few hundred thousand different (stored value, weight, pixel colour, colour weight) combinations through the
    running averages and their quantisation (pack truncation, the .5 ties of the colour bytes).  `reachable` restricts the draw to
    states the reference's own kernels can leave behind (weights <= 128, a voxel of weight 0 holds colour 0, no raw -32768)."""
    vol = rng.integers(-32767 if reachable else -32768, 32768, (N, N, N)).astype(np.int16)
    vol[rng.random((N, N, N)) < 0.3] = 32767          # saturated free space, the common state
    col = rng.integers(0, 256, (N, N, N, 4)).astype(np.uint8)
    w = rng.integers(0, 129 if reachable else 256, (N, N, N))
    w[rng.random((N, N, N)) < 0.2] = 128
    col[..., 3] = w
    if reachable:
        col[w == 0] = 0
    else:
        col[rng.random((N, N, N)) < 0.05, :3] = 0     # the "stored colour is black" branch of tsdf_volume.cu:623
    return vol, col


def perturbed_maps(rng, v, n, frac=0.2):
    """Vertex / normal maps (3 planes of rows) with a fraction of the pixels replaced by random vertices (metres away from their
    neighbours), random un-normalised normals and NaN holes: the correspondence search of the ICP kernels then meets every branch of its
    validity tests next to every other."""
    rows = v.shape[0] // 3
    v, n = v.copy(), n.copy()
    for arr, scale in ((v, 3.0), (n, 1.5)):
        m = rng.random((rows, v.shape[1])) < frac
        for k in range(3):
            arr[k * rows:(k + 1) * rows][m] = rng.uniform(-scale, scale, int(m.sum())).astype(np.float32)
    hole = rng.random((rows, v.shape[1])) < frac / 4
    v[:rows][hole] = np.nan
    hole = rng.random((rows, v.shape[1])) < frac / 4
    n[:rows][hole] = np.nan
    return v, n
