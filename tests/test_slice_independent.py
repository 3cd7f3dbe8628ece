"""CPU only.  An INDEPENDENT check of the (f2) stage: the oracle's kto_slice_process restates PCL 1.7's VoxelGrid and
NormalEstimation (backend/CloudSliceProcessor.cpp:87-163; PCL itself is absent and stays UNPINNED), and since round 4 the HIP kernels and the
oracle share one hand-written trig, so tests/test_slice_process.py proves HIP == restatement and nothing more.  Here the same stage is written
a second time with different tools and no shared code: numpy leaf binning, scipy.spatial.cKDTree for the 20 nearest neighbours,
numpy.linalg.eigh (float64 LAPACK) on a float64 two-pass covariance, flip towards the viewpoint -- and compared with the oracle on slices
the oracle's own tracker extracts.  A wrong coefficient of the trig, a wrong tie rule, a transposed covariance or a wrong leaf order can
no longer pass on both sides.
What it can and cannot show: leaf membership and order are exact; centroids agree to float rounding; normals and curvature agree up to
the error the reference's OWN arithmetic has -- pcl::computeMeanAndCovarianceMatrix accumulates raw float coordinates in one pass
(E[x x^T] - m m^T with |x| of metres and a spread of centimetres), which perturbs the covariance by ~k eps |x|^2 and the normal by that over
the eigen-gap.  The bound below is computed per point from exactly that; the medians sit far under it."""
import numpy as np
import pytest
from scipy.spatial import cKDTree


def _slices(oracle_mod):
    """Shift slabs (thin strips: the 20 neighbours of an edge point lie along the strip) + the final cloud, extracted by the oracle's tracker
    from 150 frames of a small crab-walk (the first slabs that carry surface leave the volume after ~130 frames)."""
    from kintinuous_amd import synth
    from oracle.oracle import OTrackerConfig, OracleTracker
    cam = synth.Camera.small(160, 120)
    scene = synth.Scene("wall")
    traj = synth.crabwalk_trajectory(420)
    otr = OracleTracker(OTrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 7.0, 3, 2, 0, 0, 0, 0, 0, 0))
    for k in range(150):
        d, rgb = synth.render(scene, cam, *traj[k])
        otr.process_frame(d, rgb, 33333 * k)
    otr.finalise()
    out = [otr.slice(i)[0] for i in range(otr.num_slices())]
    otr.close()
    return out, 7.0 / 96


def independent_stage(points, cull, leaf, k=20):
    """CloudSliceProcessor.cpp:87-163 from its description, float64 wherever the definition allows: returns (leaf keys in output order,
    members per leaf, centroid f64, colour mean f64, normal f64, curvature f64, eigen-gap, |x|^2 scale)."""
    pts = points[points["bgra"][:, 3] >= cull] if cull > 0 else points            # :99-117
    xyz32 = pts["xyz"].astype(np.float32)
    # pcl::VoxelGrid::applyFilter: leaf index = floor(coordinate * inverse_leaf_size) in FLOAT (that product defines membership), relative
    # to the floor of the cloud's minimum; output in ascending linear index
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(xyz32 * inv).astype(np.int64)
    lo = np.floor(xyz32.min(axis=0) * inv).astype(np.int64)
    hi = np.floor(xyz32.max(axis=0) * inv).astype(np.int64)
    ijk -= lo
    div = hi - lo + 1
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq, inverse, counts = np.unique(key, return_inverse=True, return_counts=True)
    cen = np.zeros((len(uniq), 3))
    col = np.zeros((len(uniq), 3))
    np.add.at(cen, inverse, pts["xyz"].astype(np.float64))
    np.add.at(col, inverse, pts["bgra"][:, :3].astype(np.float64))
    cen /= counts[:, None]
    col /= counts[:, None]
    # pcl::NormalEstimation, k nearest neighbours of every down-sampled point (itself included) in the down-sampled FLOAT cloud
    cloud = cen.astype(np.float32).astype(np.float64)
    n = len(cloud)
    normal = np.full((n, 3), np.nan)
    curv = np.full(n, np.nan)
    gap = np.full(n, np.nan)
    scale = np.full(n, np.nan)
    if n >= 3:
        kk = min(k, n)
        _, nbr = cKDTree(cloud).query(cloud, k=kk)
        nb = cloud[nbr]                                       # [n, kk, 3]
        mean = nb.mean(axis=1, keepdims=True)
        dlt = nb - mean
        cov = np.einsum("nki,nkj->nij", dlt, dlt) / kk        # population covariance, two-pass
        w, v = np.linalg.eigh(cov)                            # ascending
        nrm = v[:, :, 0]
        flip = (nrm * (0.0 - cloud)).sum(axis=1) < 0          # flipNormalTowardsViewpoint, sensor origin (0, 0, 0)
        nrm[flip] *= -1
        normal, curv = nrm, np.abs(w[:, 0] / w.sum(axis=1))   # solvePlaneParameters: |lambda_0 / trace|
        gap = w[:, 1] - w[:, 0]
        scale = (nb ** 2).sum(axis=2).max(axis=1)
    return uniq, counts, cen, col, normal, curv, gap, scale


def test_oracle_stage_against_an_independent_float64_reference(oracle_mod):
    slices, leaf = _slices(oracle_mod)
    slices = sorted([s for s in slices if len(s) > 100], key=len)
    assert len(slices) >= 3 and len(slices[-1]) > 2000, [len(s) for s in slices]   # two slabs and the final cloud
    worst_n, worst_c, total = 0.0, 0.0, 0
    for s in slices[-3:]:
        for cull in (0, 2):
            got = oracle_mod.slice_process(s, cull, leaf)
            keys, counts, cen, col, normal, curv, gap, scale = independent_stage(s, cull, leaf)
            # leaf count, membership and order: exact.  (Membership: a centroid within float rounding of the mean of exactly the
            # points that fall into the leaf; a point assigned to the wrong leaf would move it by a fraction of the leaf size.)
            assert len(got) == len(keys)
            ulp = np.spacing(np.abs(cen).astype(np.float32)).astype(np.float64)
            assert (np.abs(got["xyz"].astype(np.float64) - cen) <= (counts[:, None] + 1) * ulp).all()
            assert (np.abs(got["xyz"].astype(np.float64) - cen).max(axis=1) < 1e-3 * leaf).all()
            # colour: the float mean of the members, truncated (within 1 of the float64 mean's floor at rounding edges)
            assert (np.abs(got["bgra"][:, :3].astype(np.float64) - np.floor(col)) <= 1).all()
            ok = ~np.isnan(got["normal"]).any(axis=1)
            assert ok.all() == (len(keys) >= 3)
            if not ok.any():
                continue
            gn = got["normal"].astype(np.float64)
            assert np.abs(np.linalg.norm(gn, axis=1) - 1).max() < 1e-5
            assert ((gn * -got["xyz"]).sum(axis=1) >= 0).all()                       # towards the sensor origin
            # error of the reference's own arithmetic: single-pass float covariance (see the module docstring).  Measured on these slices:
            # median 4e-5 .. 7e-5, maximum 7e-4 = 0.55 of this bound
            eps = 2.0 ** -24
            tol = 5e-5 + 16 * eps * scale / np.maximum(gap, 1e-12)
            dn = np.linalg.norm(gn - normal, axis=1)
            dflip = np.linalg.norm(gn + normal, axis=1)        # a normal perpendicular to the view ray may flip either way
            grazing = np.abs((normal * cloud_dir(cen)).sum(axis=1)) < 10 * tol
            d = np.where(grazing, np.minimum(dn, dflip), dn)
            assert (d <= tol).all(), float((d / tol).max())       # every point: the bound adapts to the point's eigen-gap
            assert np.median(d) < 2e-4 and d.max() < 2e-3
            dc = np.abs(got["curvature"].astype(np.float64) - curv)
            tol_c = 1e-4 * curv + 16 * eps * scale / np.maximum(w_trace(s, cull, leaf, cen), 1e-12) + 1e-6   # (measured: median 1e-5 .. 3e-5, maximum 2e-4)
            assert (dc <= tol_c).all(), float((dc / tol_c).max())
            assert np.median(dc) < 1e-4 and dc.max() < 6e-4
            worst_n, worst_c, total = max(worst_n, float(np.median(d))), max(worst_c, float(np.median(dc))), total + int(ok.sum())
    assert total > 1000


def cloud_dir(cen):
    """unit vectors from the sensor origin to the points"""
    c = cen.astype(np.float64)
    return c / np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-12)


def w_trace(s, cull, leaf, cen, k=20):
    """trace of the k-neighbourhood covariance of every down-sampled point (the denominator of the curvature)"""
    cloud = cen.astype(np.float32).astype(np.float64)
    kk = min(k, len(cloud))
    _, nbr = cKDTree(cloud).query(cloud, k=kk)
    nb = cloud[nbr]
    return ((nb - nb.mean(axis=1, keepdims=True)) ** 2).sum(axis=(1, 2)) / kk
