"""GPU: the SECOND arithmetic contract of the voxel kernel, "survey-8c" (kt_tsdf23_tol_kernel, csrc/kt_volume.hip; VERDICT r4 item 2 iii).

The bit-exact kernel (kt_tsdf23_lean_kernel) is the library's default and what every other parity test and bench.py's headline run.  The
tolerant kernel drops the correctly rounded square root inside the truncation band (v_sqrt_f32, <= 1 ulp), the Markstein correction of the
running average and the residual test of the colour blend; it keeps the projection's exact reciprocal, so a voxel reads the SAME pixel, and
every predicate, weight and store rule.  What that changes is COUNTED here, at the sizes of BASELINE configs 2, 3 and 5, in two ways:

1. ONE CALL ON IDENTICAL INPUTS (`kernel_report`): a volume state grown by the bit-exact kernel over a stretch of the sequence, then the next
   frame fused into copies of that state by both kernels.  This is the level at which SURVEY.md 8(c)'s parity policy and north_star's
   "1e-4 on floats, bit-exact on voxel indices" are statements about the kernel:
     * weights: identical, except where the update predicate `sdf >= -trunc` sits within the root's last bit of its edge -- a voxel then is
       updated under one contract and not under the other (the reference itself, built --prec-sqrt=false, sits on the other side of such
       edges).  Counted ("edge voxels"), bar 1e-6 of the updated voxels; those voxels are excluded from the two bars below;
     * tsdf shorts: every difference exactly +-1 (3e-5 of the value range: inside north_star's 1e-4); the fraction is REPORTED and held to a
       measured bar, not to SURVEY 8(c)'s expectation of 1e-5 -- pack_tsdf truncates F x 32767, so a relative error e moves it across an
       integer with probability ~32767 e |F| (2e-3 |F| at one ulp): no arithmetic that is off by a single ulp can meet 1e-5;
     * colour bytes: every difference exactly +-1, at rn() ties -- with the usual colour weight of exactly 2 the quotient (c W + 2 p) / (W + 2)
       is a ratio of small integers and sits EXACTLY on .5 for one updated byte in a few thousand, where the approximate quotient lands on
       either side ("identical under the same caveat for rn ties"); measured 7e-5 .. 7e-4 of the updated bytes, bar 2e-3.
2. WHOLE RUNS (`report`): the tracker under the tolerant contract against the oracle.  Poses stay within 1e-4 (north_star's bar; measured
   5e-6 .. 2e-5), shift decisions and slice counts identical.  The VOLUMES do not stay within +-1: a pose that differs in its sixth digit
   moves voxels across pixel boundaries, so a few per cent of the touched voxels see another depth sample after 30 frames -- any contract
   short of bit-exactness diverges this way, which is the practical argument for the bit-exact default.  Reported, loosely bounded.

`scripts/tol_contract_report.py` prints both reports into profiles/r05_tol_contract.jsonl."""
import numpy as np
import pytest

from test_gpu_tracker import _cfgs

pytestmark = pytest.mark.gpu


class contract:
    """with contract(tol): the lean voxel kernel runs under the survey-8c contract inside the block"""

    def __init__(self, tol):
        self.tol = tol

    def __enter__(self):
        from kintinuous_amd import abi
        abi._chk(abi.lib().kt_debug_tsdf_lean(1))
        abi._chk(abi.lib().kt_debug_tsdf_contract(1 if self.tol else 0))
        want = b"kt_tsdf23_tol_kernel" if self.tol else b"kt_tsdf23_lean_kernel"
        assert abi.lib().kt_debug_tsdf_kernel() == want

    def __exit__(self, *exc):
        from kintinuous_amd import abi
        abi.lib().kt_debug_tsdf_contract(-1)
        abi.lib().kt_debug_tsdf_lean(-1)


def kernel_report(ctx, config, N, grow, cam=None, wrap=(0, 0, 0)):
    """Volume state = frames 0 .. grow-1 of `config` fused at their ground-truth poses by the BIT-EXACT kernel; then frame `grow` fused into
    two copies of that state, once per contract.  Returns the counted differences of that one call."""
    from hip_kernels import HipKernels
    from kintinuous_amd import abi, synth
    from kintinuous_amd.abi import Intr
    from oracle import oracle as O          # host-side 3x3 inverse only (the reference computes it with Eigen before the launch)
    from oracle.oracle import OIntr
    cam = cam or synth.Camera()
    _, frames, traj, kw = synth.sequence(config, grow + 1, cam)
    size = float(kw["volume_size"])
    trunc = max(0.06 if size == 6.0 else size / 100.0, 2.1 * size / N)
    H = HipKernels(ctx)
    intr, ointr = Intr(cam.fx, cam.fy, cam.cx, cam.cy), OIntr(cam.fx, cam.fy, cam.cx, cam.cy)
    nvox = N * N * N
    dv, dc = ctx.zeros(nvox * 2), ctx.zeros(nvox * 4)
    sc = ctx.zeros(cam.rows * cam.cols * 4)
    basis = np.float32(size / 2)
    static = bool(kw.get("static_mode"))

    def fuse(k, vol, col):
        d, c = frames[k]
        R, cc = np.asarray(traj[k][0], np.float32), np.asarray(traj[k][1], np.float32)
        t = (cc + basis).astype(np.float32)
        if static:   # -sm: the camera 0.45 m outside the near face (KintinuousTracker.cpp:103-106)
            t[2] = np.float32(size * 0.5) - np.float32(size * 0.5 + 0.45) + cc[2]
        n = H.create_nmap(H.create_vmap(ointr, H.bilateral_filter(d)))
        ctx.integrate_tsdf(ctx.upload(np.ascontiguousarray(d, np.uint16)), cam.cols, cam.rows, intr, [size] * 3, O.mat33_inverse(R), t, trunc, vol, sc,
                           list(wrap), col, ctx.upload(np.ascontiguousarray(c, np.uint8)), ctx.upload(np.ascontiguousarray(n, np.float32)), True, N)

    with contract(False):
        for k in range(grow):
            fuse(k, dv, dc)
        ctx.sync()
        v0, c0 = ctx.download(dv, np.int16, (nvox,)), ctx.download(dc, np.uint8, (nvox, 4))
        fuse(grow, dv, dc)
        ctx.sync()
        ve, ce = ctx.download(dv, np.int16, (nvox,)), ctx.download(dc, np.uint8, (nvox, 4))
    dv2, dc2 = ctx.upload(v0), ctx.upload(c0)
    with contract(True):
        fuse(grow, dv2, dc2)
        ctx.sync()
        vt, ct = ctx.download(dv2, np.int16, (nvox,)), ctx.download(dc2, np.uint8, (nvox, 4))
    changed = (ve != v0) | (ce != c0).any(axis=1)
    out = {"config": config, "N": N, "state_frames": grow, "voxels_changed_by_the_call": int(changed.sum())}
    edge = ce[:, 3] != ct[:, 3]                                   # the update predicate fell differently
    out["edge_voxels"] = int(edge.sum())
    keep = ~edge
    dvv = (vt.astype(np.int32) - ve.astype(np.int32))[keep]
    out["tsdf_diff_voxels"] = int((dvv != 0).sum())
    out["tsdf_diff_max"] = int(np.abs(dvv).max())
    dcc = (ct[:, :3].astype(np.int16) - ce[:, :3].astype(np.int16))[keep]
    out["colour_diff_bytes"] = int((dcc != 0).sum())
    out["colour_diff_max"] = int(np.abs(dcc).max())
    n = max(1, out["voxels_changed_by_the_call"])
    out["edge_fraction"] = out["edge_voxels"] / n
    out["tsdf_diff_fraction"] = out["tsdf_diff_voxels"] / n
    out["colour_diff_fraction"] = out["colour_diff_bytes"] / (3 * n)
    return out


def report(ctx, config, nframes, N, cam=None):
    """`config` through the HIP tracker under the survey-8c contract and through the oracle: poses, decisions, and how far the volumes drift."""
    from kintinuous_amd import abi, synth
    from oracle import oracle
    cam = cam or synth.Camera()
    _, frames, traj, kw = synth.sequence(config, nframes, cam)
    g, o = _cfgs(cam, N, **kw)
    out = {"config": config, "frames": nframes, "N": N}
    with contract(True):
        trk, otr = abi.Tracker(ctx, g), oracle.OracleTracker(o)
        dev = [(ctx.upload(d), ctx.upload(c)) for d, c in frames]
        for k in range(len(frames)):
            if k + 1 < len(frames):
                trk.prefetch_frame(*dev[k + 1])
            trk.process_frame(dev[k][0], dev[k][1], 33333 * k)
        for k, (d, c) in enumerate(frames):
            otr.process_frame(d, c, 33333 * k)
        assert trk.num_poses() == otr.num_poses() == len(frames)
        rel = 0.0
        for i in range(len(frames)):
            p, op = np.asarray(trk.dense_pose(i)[1]).reshape(4, 4), np.asarray(otr.dense_pose(i)[1]).reshape(4, 4)
            rel = max(rel, float(np.linalg.norm(p[:3, :3] - op[:3, :3])), float(np.linalg.norm(p[:3, 3] - op[:3, 3]) / max(1.0, np.linalg.norm(op[:3, 3]))))
        out["pose_rel_max"] = rel
        out["wraps_equal"] = bool(np.array_equal(trk.voxel_wrap(), otr.voxel_wrap()))
        out["wrap"] = [int(x) for x in otr.voxel_wrap()]
        v, ov = trk.volume(), otr.volume()
        c, oc = trk.color_volume(), otr.color_volume()
        touched = int((oc[..., 3] != 0).sum())
        dv = np.abs(v.astype(np.int32) - ov.astype(np.int32))
        out["touched_voxels"] = touched
        out["tsdf_diff_fraction_of_touched"] = int((dv != 0).sum()) / max(1, touched)
        out["tsdf_diff_gt1_fraction_of_touched"] = int((dv > 1).sum()) / max(1, touched)
        out["tsdf_mean_abs_diff_shorts_over_touched"] = float(dv.sum()) / max(1, touched)
        del dv, v, ov
        out["weight_diff_fraction_of_touched"] = int((c[..., 3] != oc[..., 3]).sum()) / max(1, touched)
        del c, oc
        out["slices"] = [trk.num_slices(), otr.num_slices()]
        out["slice_sizes"] = [[len(trk.slice(i)[0]), len(otr.slice(i)[0])] for i in range(min(trk.num_slices(), otr.num_slices()))]
        trk.close(); otr.close()
    return out


def _check_kernel(r, tsdf_bar):
    print(r)
    assert r["voxels_changed_by_the_call"] > 100000, r
    assert r["edge_fraction"] <= 1e-6, r
    assert r["tsdf_diff_max"] <= 1 and r["colour_diff_max"] <= 1, r          # outside the edge voxels every difference is exactly one
    assert r["tsdf_diff_fraction"] <= tsdf_bar, r
    assert r["colour_diff_fraction"] <= 2e-3, r


def _check_run(r):
    print(r)
    assert r["pose_rel_max"] <= 1e-4, r
    assert r["wraps_equal"] and r["slices"][0] == r["slices"][1], r
    for a, b in r["slice_sizes"]:
        assert abs(a - b) <= 0.02 * max(a, b, 50), r
    # drift of the volumes behind a pose that differs in its sixth digit (see the module docstring): bounded loosely, reported exactly
    assert r["tsdf_diff_gt1_fraction_of_touched"] <= 0.10 and r["weight_diff_fraction_of_touched"] <= 0.01, r


# kernel-level bars: ~3x the measured fractions (tsdf: 2e-4 .. 7.5e-4) (profiles/r05_tol_contract.jsonl), so that a contract that silently got looser fails
def test_config2_orbit512_one_call_under_the_survey8c_contract(ctx, oracle_mod):
    _check_kernel(kernel_report(ctx, "orbit", 512, 12, wrap=(37, 501, 130)), tsdf_bar=2.5e-3)


def test_config3_crabwalk512_one_call_under_the_survey8c_contract(ctx, oracle_mod):
    _check_kernel(kernel_report(ctx, "crabwalk", 512, 12, wrap=(5, 0, 500)), tsdf_bar=2.5e-3)


def test_config5_farwall768_one_call_under_the_survey8c_contract(ctx, oracle_mod):
    from kintinuous_amd import synth
    cam = synth.Camera(1280, 960, 2 * synth.FX, 2 * synth.FY, 2 * synth.CX, 2 * synth.CY)
    _check_kernel(kernel_report(ctx, "farwall", 768, 1, cam), tsdf_bar=2.5e-3)


def test_config2_orbit512_run_under_the_survey8c_contract(ctx, oracle_mod):
    _check_run(report(ctx, "orbit", 34, 512))


def test_config3_crabwalk512_run_under_the_survey8c_contract(ctx, oracle_mod):
    r = report(ctx, "crabwalk", 29, 512)
    assert r["wrap"][0] >= 28 and r["slices"][1] >= 2
    _check_run(r)


def test_the_default_contract_is_bit_exact(ctx):
    """nothing selects the tolerant kernel unless asked: the library's default, after the hooks above have been released, is the lean kernel"""
    import os
    from kintinuous_amd import abi
    if os.environ.get("KT_TSDF_CONTRACT") or os.environ.get("KT_TSDF_LEAN") == "0":
        pytest.skip("the environment selects a kernel")
    assert abi.lib().kt_debug_tsdf_kernel() == b"kt_tsdf23_lean_kernel"
