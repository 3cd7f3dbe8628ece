"""GPU parity: image-side kernels (SURVEY 8a rows a1-a5, a13, a10 helpers) through the C-ABI vs the oracle.  Bit-exact."""
import numpy as np
import pytest

from conftest import random_rotation

pytestmark = pytest.mark.gpu


def _depth_cases(small_scene):
    cam, frames, _ = small_scene
    rng = np.random.default_rng(11)
    d0 = frames[0][0]
    noisy = d0.astype(np.int32) + rng.integers(-40, 40, d0.shape)
    noisy[d0 == 0] = 0
    holes = noisy.copy()
    holes[rng.uniform(size=d0.shape) < 0.1] = 0
    holes[:, -3:] = 0
    full = rng.integers(0, 65536, d0.shape)  # full uint16 range incl. > 32767
    return cam, [d0, np.clip(noisy, 0, 65535).astype(np.uint16), np.clip(holes, 0, 65535).astype(np.uint16), full.astype(np.uint16),
                 np.zeros_like(d0)]


def test_bilateral_and_pyramid(ctx, oracle_mod, small_scene):
    cam, cases = _depth_cases(small_scene)
    rows, cols = cam.rows, cam.cols
    for i, d in enumerate(cases):
        ref = oracle_mod.bilateral_filter(d)
        dst = ctx.empty(d.nbytes)
        ctx.bilateral_filter(ctx.upload(d), dst, cols, rows)
        got = ctx.download(dst, np.uint16, d.shape)
        assert np.array_equal(ref, got), f"case {i}: {(ref != got).sum()} mismatches"
        cur, dcur, c, r = ref, dst, cols, rows
        for l in range(3):
            nxt = oracle_mod.pyr_down(cur)
            dn = ctx.empty(nxt.nbytes)
            ctx.pyr_down(dcur, c, r, dn)
            assert np.array_equal(nxt, ctx.download(dn, np.uint16, nxt.shape)), f"case {i} level {l + 1}"
            cur, dcur, c, r = nxt, dn, c // 2, r // 2


def test_bilateral_vga_ragged(ctx, oracle_mod):
    """Full 640x480 frame and a ragged size that is not a multiple of the 16x16 tile."""
    from kintinuous_amd import synth
    for (c, r) in ((640, 480), (200, 72)):
        cam = synth.Camera.small(c, r)
        d, _ = synth.render(synth.Scene("room"), cam, np.eye(3), np.zeros(3))
        ref = oracle_mod.bilateral_filter(d)
        dst = ctx.empty(d.nbytes)
        ctx.bilateral_filter(ctx.upload(d), dst, c, r)
        assert np.array_equal(ref, ctx.download(dst, np.uint16, d.shape))


def test_vmap_nmap_transform_resize(ctx, oracle_mod, small_scene):
    from kintinuous_amd.abi import Intr
    from oracle.oracle import OIntr
    cam, cases = _depth_cases(small_scene)
    rows, cols = cam.rows, cam.cols
    rng = np.random.default_rng(5)
    for i, d in enumerate(cases[:3]):
        for level in (0, 1):
            dd = d if level == 0 else oracle_mod.pyr_down(d)
            r, c = dd.shape
            oi, gi = OIntr(cam.fx, cam.fy, cam.cx, cam.cy).level(level), Intr(cam.fx, cam.fy, cam.cx, cam.cy).level(level)
            # stale-plane semantics: outputs start from the same pre-filled buffers on both sides
            pre = rng.uniform(-1, 1, (3 * r, c)).astype(np.float32)
            vref = oracle_mod.create_vmap(oi, dd, out=pre.copy())
            dv = ctx.upload(pre)
            ctx.create_vmap(gi, ctx.upload(dd), c, r, dv)
            assert np.array_equal(vref.view(np.uint32), ctx.download(dv, np.float32, vref.shape).view(np.uint32))
            nref = oracle_mod.create_nmap(vref, out=pre.copy())
            dn = ctx.upload(pre)
            ctx.create_nmap(dv, c, r, dn)
            assert np.array_equal(nref.view(np.uint32), ctx.download(dn, np.float32, nref.shape).view(np.uint32))
            R, t = random_rotation(rng, 1.0), rng.uniform(-3, 3, 3).astype(np.float32)
            tv, tn = oracle_mod.transform_maps(vref, nref, R, t, vout=pre.copy(), nout=pre.copy())
            dtv, dtn = ctx.upload(pre), ctx.upload(pre)
            ctx.transform_maps(dv, dn, c, r, R, t, dtv, dtn)
            assert np.array_equal(tv.view(np.uint32), ctx.download(dtv, np.float32, tv.shape).view(np.uint32))
            assert np.array_equal(tn.view(np.uint32), ctx.download(dtn, np.float32, tn.shape).view(np.uint32))
            pre2 = rng.uniform(-1, 1, (3 * (r // 2), c // 2)).astype(np.float32)
            rv = oracle_mod.resize_map(tv, False, out=pre2.copy())
            rn = oracle_mod.resize_map(tn, True, out=pre2.copy())
            drv, drn = ctx.upload(pre2), ctx.upload(pre2)
            ctx.resize_vmap(dtv, c, r, drv)
            ctx.resize_nmap(dtn, c, r, drn)
            assert np.array_equal(rv.view(np.uint32), ctx.download(drv, np.float32, rv.shape).view(np.uint32))
            assert np.array_equal(rn.view(np.uint32), ctx.download(drn, np.float32, rn.shape).view(np.uint32))


def test_rgbd_pyramids(ctx, oracle_mod, small_scene):
    cam, frames, _ = small_scene
    rows, cols = cam.rows, cam.cols
    depth, rgb = frames[1]
    rng = np.random.default_rng(2)
    depth = depth.copy()
    depth[rng.uniform(size=depth.shape) < 0.05] = 0
    depth[10:20, 10:30] = 7000  # beyond the 6 m cut-off
    dm = oracle_mod.depth_to_metres(depth, 6000)
    g = ctx.empty(dm.nbytes)
    ctx.depth_to_metres(ctx.upload(depth), g, cols, rows, 6000)
    assert np.array_equal(dm.view(np.uint32), ctx.download(g, np.float32, dm.shape).view(np.uint32))
    inten = oracle_mod.bgr_to_intensity(rgb)
    gi = ctx.empty(inten.nbytes)
    ctx.bgr_to_intensity(ctx.upload(rgb), gi, cols, rows)
    assert np.array_equal(inten, ctx.download(gi, np.uint8, inten.shape))
    c, r, curd, curi, gd, gim = cols, rows, dm, inten, g, gi
    for l in range(3):
        nd, ni = oracle_mod.pyr_down_gauss_f32(curd), oracle_mod.pyr_down_gauss_u8(curi)
        gnd, gni = ctx.empty(nd.nbytes), ctx.empty(ni.nbytes)
        ctx.pyr_down_gauss_f32(gd, c, r, gnd)
        ctx.pyr_down_gauss_u8(gim, c, r, gni)
        assert np.array_equal(nd.view(np.uint32), ctx.download(gnd, np.float32, nd.shape).view(np.uint32)), f"depth level {l + 1}"
        assert np.array_equal(ni, ctx.download(gni, np.uint8, ni.shape)), f"intensity level {l + 1}"
        c, r, curd, curi, gd, gim = c // 2, r // 2, nd, ni, gnd, gni
    dx, dy = oracle_mod.derivative_images(inten)
    gdx, gdy = ctx.empty(dx.nbytes), ctx.empty(dy.nbytes)
    ctx.derivative_images(gi, cols, rows, gdx, gdy)
    assert np.array_equal(dx, ctx.download(gdx, np.int16, dx.shape))
    assert np.array_equal(dy, ctx.download(gdy, np.int16, dy.shape))
    for level in (0, 2):
        src = dm if level == 0 else oracle_mod.pyr_down_gauss_f32(oracle_mod.pyr_down_gauss_f32(dm))
        rr, cc = src.shape
        cl = oracle_mod.project_to_cloud(src, cam.fx, cam.fy, cam.cx, cam.cy, level)
        gc = ctx.empty(cl.nbytes)
        ctx.project_to_cloud(ctx.upload(src), cc, rr, gc, cam.fx, cam.fy, cam.cx, cam.cy, level)
        assert np.array_equal(cl.view(np.uint32), ctx.download(gc, np.float32, cl.shape).view(np.uint32))


@pytest.mark.parametrize("size", [(160, 120), (640, 480), (200, 72)])
def test_build_pyramid_fused(ctx, oracle_mod, size):
    """kt_build_pyramid (one launch) == pyrDown x3 + createVMap x4 + createNMap x4 of the oracle, bit for bit,
    including the stale-plane semantics of invalid pixels."""
    from kintinuous_amd import synth
    from kintinuous_amd.abi import Intr
    from oracle.oracle import OIntr
    cols, rows = size
    cam = synth.Camera.small(cols, rows)
    d, _ = synth.render(synth.Scene("room"), cam, np.eye(3), np.zeros(3))
    rng = np.random.default_rng(cols)
    d = d.copy()
    d[rng.uniform(size=d.shape) < 0.08] = 0
    d0 = oracle_mod.bilateral_filter(d)
    depths = [d0]
    for l in range(3):
        depths.append(oracle_mod.pyr_down(depths[-1]))
    pre = [rng.uniform(-1, 1, (3 * (rows >> l), cols >> l)).astype(np.float32) for l in range(4)]
    vref = [oracle_mod.create_vmap(OIntr(cam.fx, cam.fy, cam.cx, cam.cy).level(l), depths[l], out=pre[l].copy()) for l in range(4)]
    nref = [oracle_mod.create_nmap(vref[l], out=pre[l].copy()) for l in range(4)]
    gd = [ctx.zeros(depths[l].nbytes) for l in range(1, 4)]
    gv = [ctx.upload(pre[l]) for l in range(4)]
    gn = [ctx.upload(pre[l]) for l in range(4)]
    ctx.build_pyramid(Intr(cam.fx, cam.fy, cam.cx, cam.cy), ctx.upload(d0), cols, rows, gd, gv, gn)
    ctx.sync()
    for l in range(1, 4):
        assert np.array_equal(depths[l], ctx.download(gd[l - 1], np.uint16, depths[l].shape)), f"depth level {l}"
    for l in range(4):
        assert np.array_equal(vref[l].view(np.uint32), ctx.download(gv[l], np.float32, vref[l].shape).view(np.uint32)), f"vmap level {l}"
        assert np.array_equal(nref[l].view(np.uint32), ctx.download(gn[l], np.float32, nref[l].shape).view(np.uint32)), f"nmap level {l}"


def test_generate_image_and_depth(ctx, oracle_mod, small_scene):
    """generateImage / generateDepth (image_generator.cu): the view products of KintinuousTracker::getImage / getModelDepth on the
    predicted maps of a tracked sequence, plus synthetic maps that hit every branch (NaN vertex, NaN normal, every heat-map band)."""
    from kintinuous_amd import abi
    cam, frames, _ = small_scene
    cfg = abi.TrackerConfig(cam.cols, cam.rows, 96, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 0, 0, 0, 0, 0, 0)
    trk = abi.Tracker(ctx, cfg)
    for k, (d, rgb) in enumerate(frames[:4]):
        trk.process_frame_host(d, rgb, 33333 * k)
    vmap, nmap = trk.vmap_g_prev(0), trk.nmap_g_prev(0)
    R, t, _ = trk.pose()
    trk.close()
    rows, cols = cam.rows, cam.cols
    rng = np.random.default_rng(2)
    cases = [(vmap, nmap, rng.integers(0, 256, (rows, cols, 4), dtype=np.uint8))]
    v2 = rng.uniform(-3, 9, (3 * rows, cols)).astype(np.float32)
    n2 = rng.normal(size=(3 * rows, cols)).astype(np.float32)
    v2[:rows][rng.uniform(size=(rows, cols)) < 0.1] = np.nan
    n2[:rows][rng.uniform(size=(rows, cols)) < 0.1] = np.nan
    c2 = rng.integers(0, 256, (rows, cols, 4), dtype=np.uint8)
    c2[..., 3] = np.arange(rows * cols).reshape(rows, cols) % 160          # heat value from 0 to beyond 1 (weight / 128)
    cases.append((v2, n2, c2))
    light = np.array([-18.0, -18.0, -18.0], np.float32)                    # getImage: volume size * -3
    Rinv = oracle_mod.mat33_inverse(R)
    for i, (v, n, c) in enumerate(cases):
        ref_img, ref_col = oracle_mod.generate_image(v, n, c, light)
        ref_dep = oracle_mod.generate_depth(Rinv, t, v, n)
        dv, dn, dc = ctx.upload(v), ctx.upload(n), ctx.upload(c)
        di, dcol, dd = ctx.empty(rows * cols * 3), ctx.empty(rows * cols * 3), ctx.empty(rows * cols * 2)
        ctx.generate_image(dv, dn, dc, cols, rows, light, 1, di, dcol)
        ctx.generate_depth(Rinv, t, dv, dn, cols, rows, dd)
        assert np.array_equal(ref_img, ctx.download(di, np.uint8, (rows, cols, 3))), i
        assert np.array_equal(ref_col, ctx.download(dcol, np.uint8, (rows, cols, 3))), i
        assert np.array_equal(ref_dep, ctx.download(dd, np.uint16, (rows, cols))), i
    # on the tracked maps the rendered depth is the input depth seen again (the model of frame 3, within the volume's resolution)
    hit = np.isfinite(vmap[:rows]) & np.isfinite(nmap[:rows]) & (frames[3][0] > 0)
    assert hit.mean() > 0.5
    err = np.abs(oracle_mod.generate_depth(Rinv, t, vmap, nmap).astype(int) - frames[3][0].astype(int))[hit]
    assert np.median(err) < 30
