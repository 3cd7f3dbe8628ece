"""The Gauss-Newton tail of the reduction kernels in its two device forms: one thread (the restatement of Eigen's LDLT / cv::Rodrigues /
the Isometry3f update that the oracle shares, csrc/kt_track.hpp) against the lane-parallel form the kernels run
(kt_solve_and_update_wave).  Every output bit must agree, also where the pivot order hangs on ties, for singular and indefinite systems,
zeros and NaNs."""
import ctypes as C

import numpy as np
import pytest

from kintinuous_amd import abi

pytestmark = pytest.mark.gpu

CASE = np.dtype([("packed", np.float32, 32), ("packed2", np.float32, 32), ("resultRt", np.float64, 16), ("posef", np.float32, 12),
                 ("joint", np.int32), ("pad", np.int32, 3)])


def _slot(i, j):   # reduce.cu:401-418: rows i = 0..5, columns j = i..6
    return i * 7 - (i * (i - 1)) // 2 + (j - i)


def _pack(A, b):
    out = np.zeros(32, np.float32)
    for i in range(6):
        for j in range(i, 6):
            out[_slot(i, j)] = A[i, j]
        out[_slot(i, 6)] = b[i]
    return out


def _rot(rng, scale):
    w = rng.normal(size=3) * scale
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * (K @ K) if th > 0 else np.eye(3)


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    cs = np.zeros(n, CASE)
    for k in range(n):
        kind = k % 12
        m = int(rng.integers(3, 400))
        J = rng.normal(size=(m, 6)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-2, 2))
        r = rng.normal(size=m).astype(np.float32) * np.float32(10.0 ** rng.uniform(-4, 0))
        A = (J.T @ J).astype(np.float32)
        b = (J.T @ r).astype(np.float32)
        if kind == 1:     # ties on the diagonal: the pivot order is decided by the swap history
            d = rng.permutation(np.array([3.0, 3.0, 5.0, 1.0, 3.0, 5.0], np.float32))
            A = (A * np.float32(1e-3)).astype(np.float32)
            A[np.arange(6), np.arange(6)] = d
        elif kind == 2:   # all diagonal entries equal
            A[np.arange(6), np.arange(6)] = np.float32(7.0)
        elif kind == 3:   # rank deficient: rows / columns of zeros (no correspondences constrain them)
            z = rng.choice(6, size=int(rng.integers(1, 5)), replace=False)
            A[z, :] = 0; A[:, z] = 0; b[z] = 0
        elif kind == 4:   # nothing at all
            A[:] = 0; b[:] = 0
        elif kind == 5:   # rank 1 / rank 2
            J2 = J[: int(rng.integers(1, 3))]
            A = (J2.T @ J2).astype(np.float32); b = (J2.T @ r[: len(J2)]).astype(np.float32)
        elif kind == 6:   # indefinite
            A = (A - np.float32(0.5) * np.diag(np.diag(A))).astype(np.float32)
            A[2, 2] = -A[2, 2]
        elif kind == 7:   # a NaN / an infinity somewhere
            A[int(rng.integers(6)), int(rng.integers(6))] = np.float32(np.nan if k % 2 else np.inf)
            A = np.triu(A) + np.triu(A, 1).T
        elif kind == 8:   # a tiny increment: Rodrigues' theta < eps branch
            b = (b * np.float32(1e-30)).astype(np.float32)
        elif kind == 9:   # a huge increment: the large-argument path of sin / cos
            b = (b * np.float32(1e6)).astype(np.float32)
        cs[k]["packed"] = _pack(A, b)
        if kind == 10 or kind == 11:   # the joint RGB-D + ICP combination
            J3 = rng.normal(size=(m, 6)).astype(np.float32)
            cs[k]["packed2"] = _pack((J3.T @ J3).astype(np.float32), (J3.T @ r).astype(np.float32))
            cs[k]["joint"] = 1
        T = np.eye(4)
        T[:3, :3] = _rot(rng, 0.05); T[:3, 3] = rng.normal(size=3) * 0.05
        cs[k]["resultRt"] = T.reshape(16)
        cs[k]["posef"][:9] = _rot(rng, 1.0).astype(np.float32).reshape(9)
        cs[k]["posef"][9:] = (rng.normal(size=3) * 3).astype(np.float32)
    return cs


def test_lane_parallel_tail_equals_the_serial_one():
    ctx = abi.Ctx(0)
    layout = (C.c_int * 5)()
    abi._chk(abi.lib().kt_debug_solve_check(ctx.h, 0, None, None, None, layout))
    size, o_rt, o_R, o_t, case_size = list(layout)
    assert case_size == CASE.itemsize
    n = 6000
    cs = _cases(n, 20240924)
    ser = np.zeros((n, size), np.uint8)
    wav = np.zeros((n, size), np.uint8)
    abi._chk(abi.lib().kt_debug_solve_check(ctx.h, n, cs.ctypes.data_as(C.c_void_p), ser.ctypes.data_as(C.c_void_p), wav.ctypes.data_as(C.c_void_p), layout))
    fields = {"resultRt": (o_rt, 128), "Rcurr": (o_R, 36), "tcurr": (o_t, 12)}
    # the serial form did something (not a buffer of zeros), and the two agree in every byte of every output
    assert np.any(ser[:, o_rt:o_rt + 128] != 0)
    for name, (off, nb) in fields.items():
        ty = np.float64 if name == "resultRt" else np.float32
        a, b = ser[:, off:off + nb].copy().view(ty), wav[:, off:off + nb].copy().view(ty)
        # a NaN is a NaN (its sign and payload are not the reference's either: x86 and gfx950 differ there); everything else bit for bit
        same = (np.isnan(a) & np.isnan(b)) | (a.view(np.uint64 if ty is np.float64 else np.uint32) == b.view(np.uint64 if ty is np.float64 else np.uint32))
        bad = np.nonzero(~np.all(same, axis=1))[0]
        assert bad.size == 0, (name, bad[:10], [int(k) % 12 for k in bad[:10]], a[bad[0]], b[bad[0]])
    # healthy systems produce finite poses (the comparison above is not NaN against NaN throughout)
    healthy = [k for k in range(n) if k % 12 == 0]
    assert np.all(np.isfinite(ser[healthy][:, o_rt:o_rt + 128].copy().view(np.float64)))
    ctx.close()
