"""One kernel-level scenario, run through any implementation M with the function names of oracle/oracle.py:
   - make_golden_ref.py runs it through oracle/_ref (the reference's own .cu sources compiled for the CPU) -> golden_ref_v1.npz;
   - tests/test_golden_ref.py runs it through the oracle (CPU) and the HIP path (GPU) and compares with the committed file.
Returns {name: array}.  Outputs whose order the reference leaves unspecified (extracted points) are sorted."""
import numpy as np


def _sorted_points(p):
    raw = np.ascontiguousarray(p).view(np.uint8).reshape(len(p), 32)
    key = np.concatenate([raw[:, :12], raw[:, 16:20]], axis=1)
    order = np.lexsort(key.T[::-1])
    return key[order]


def z_wrap_rotation(vol, col, wrap_z):
    """(k, n): rolling the STORAGE arrays by k planes along z puts the pair of logical planes with the most dz zero crossings (n of them)
    at logical (N - 1, 0)."""
    N = vol.shape[0]
    L = np.roll(vol, -wrap_z, axis=0).astype(np.int32)                 # logical z order
    ok = (np.roll(col[..., 3], -wrap_z, axis=0) != 0) & (L != 32767)
    cross = ok[:-1] & ok[1:] & (((L[:-1] > 0) & (L[1:] < 0)) | ((L[:-1] < 0) & (L[1:] > 0)))
    n = cross.reshape(N - 1, -1).sum(axis=1)
    zbest = int(np.argmax(n))
    return (N - 1 - zbest) % N, int(n[zbest])


def scenario(M, g, intr_cls):
    cols, rows = int(g["cols"]), int(g["rows"])
    fx, fy, cx, cy = [float(x) for x in g["intr"]]
    intr = intr_cls(fx, fy, cx, cy)
    out = {}
    d0, rgb0, d1, rgb1 = g["depth0"], g["rgb0"], g["depth1"], g["rgb1"]
    # the bilateral filter is the one kernel whose output depends on the __expf model (a tie of rn() can flip by the last bit of
    # a weight): it is an output here, compared with that bound, and everything downstream starts from the stored filtered frames
    out["bilateral0"] = M.bilateral_filter(d0)
    f0, f1 = g["filtered0"], g["filtered1"]
    out["pyr1"] = M.pyr_down(f0)
    v0 = M.create_vmap(intr, f0)
    n0 = M.create_nmap(v0)
    out["vmap0"], out["nmap0"] = v0, n0
    Rg = g["R_g"]
    vg, ng = M.transform_maps(v0, n0, Rg, g["t_g"])
    out["vmap_g"], out["nmap_g"] = vg, ng
    out["vmap_half"], out["nmap_half"] = M.resize_map(v0, False), M.resize_map(n0, True)
    # volume: two frames fused at two poses into wrapped storage, raycast, extraction, clears
    N, size, trunc = int(g["N"]), float(g["size"]), float(g["trunc"])
    wrap = [int(w) for w in g["wrap"]]
    vol, col = np.zeros((N, N, N), np.int16), np.zeros((N, N, N, 4), np.uint8)
    out["scaled0"] = M.integrate_tsdf(d0, intr, [size] * 3, g["Rinv0"], g["t0"], trunc, vol, wrap, col, rgb0, n0, True)
    if isinstance(out["scaled0"], tuple):
        out["scaled0"] = out["scaled0"][1]
    v1 = M.create_vmap(intr, f1)
    n1 = M.create_nmap(v1)
    M.integrate_tsdf(d1, intr, [size] * 3, g["Rinv1"], g["t1"], trunc, vol, wrap, col, rgb1, n1, True)
    out["vol"], out["col"] = vol.copy(), col.copy()
    vm = np.zeros((3 * rows, cols), np.float32)
    nm = np.zeros_like(vm)
    cm = np.zeros((rows, cols, 4), np.uint8)
    M.raycast(intr, g["R1"], g["t1"], trunc, [size] * 3, vol, vm, nm, wrap, cm, col)
    out["ray_vmap"], out["ray_nmap"], out["ray_rgb"] = vm, nm, cm[..., :3].copy()
    pts = M.extract_cloud_slice(vol, [size] * 3, 200000, wrap, col, 0, N, 0, N, 0, N, 1, [int(w) for w in g["real_wrap"]])
    out["cloud"] = _sorted_points(pts)
    pts = M.extract_cloud_slice(vol, [size] * 3, 200000, wrap, col, 0, N, 0, N, N - 13, N, 1, [int(w) for w in g["real_wrap"]])
    out["cloud_zminus"] = _sorted_points(pts)
    # extract.cu's dz neighbour of the LAST logical plane is fetched through the storage index, i.e. it wraps to logical plane 0
    # (oracle: kto_extract_cloud_slice).  The fused scene has no surface there, so the volume is rotated along z
    # until its busiest pair of planes (z, z + 1) -- the far wall -- sits at (N - 1, 0): those crossings exist only through the wrap.
    zr = z_wrap_rotation(vol, col, wrap[2])
    out["zwrap_shift_crossings"] = np.array(zr, np.int64)
    vol_r, col_r = np.roll(vol, zr[0], axis=0), np.roll(col, zr[0], axis=0)
    pts = M.extract_cloud_slice(vol_r, [size] * 3, 200000, wrap, col_r, 0, N, 0, N, 0, N, 1, [int(w) for w in g["real_wrap"]])
    out["cloud_zwrap"] = _sorted_points(pts)
    pts = M.extract_cloud_slice(vol_r, [size] * 3, 200000, wrap, col_r, 0, N, 0, N, N - 3, N, 1, [int(w) for w in g["real_wrap"]])
    out["cloud_zwrap_top"] = _sorted_points(pts)
    cv, cc = vol.copy(), col.copy()
    M.clear_volume(cv, 0, False, 20, 36)   # 17-plane X slab: the launch-geometry quirk leaves one plane
    M.clear_volume(cc, 2, True, 5, -9)
    out["clear_x"], out["clear_zc"] = cv, cc
    # ICP system of frame 1 against the raycast of the fused volume
    A, b, r = M.icp_step(g["R1"], g["t1"], v1, n1, g["Rinv0"], g["t0"], intr, vm, nm, 0.10, float(g["angle_thres"]))[:3]
    out["icp_A"], out["icp_b"], out["icp_r"] = A, b, r
    # RGB-D pieces
    dm0, dm1 = M.depth_to_metres(d0, 6000), M.depth_to_metres(d1, 6000)
    i0, i1 = M.bgr_to_intensity(rgb0), M.bgr_to_intensity(rgb1)
    dx, dy = M.derivative_images(i1)
    out["metres0"], out["intensity0"], out["dIdx1"], out["dIdy1"] = dm0, i0, dx, dy
    out["gauss_f32"], out["gauss_u8"] = M.pyr_down_gauss_f32(dm0), M.pyr_down_gauss_u8(i0)
    cloud = M.project_to_cloud(dm0, fx, fy, cx, cy, 0)
    out["cloud0"] = cloud
    corres, sigma, count = M.rgb_residual(float(g["min_scale"]), dx, dy, dm0, dm1, i0, i1, 0.07, g["kt"], g["krkinv"])
    valid = np.asarray(corres["valid"]) != 0
    out["rgb_sigma_count"] = np.array([sigma, count], np.int64)
    out["rgb_valid"] = valid
    out["rgb_zero"], out["rgb_one"], out["rgb_diff"] = corres["zero"][valid], corres["one"][valid], corres["diff"][valid]
    clean = np.zeros(corres.shape, corres.dtype)      # invalid entries are uninitialised in the reference (quirk A.19)
    for f in ("zero", "one", "diff"):
        clean[f][valid] = corres[f][valid]
    clean["valid"] = valid
    A, b = M.rgb_step(clean, float(np.sqrt(np.float32(max(count, 1)))), cloud, np.float32(fx), np.float32(fy), dx, dy, 0.125)[:2]
    out["rgb_A"], out["rgb_b"] = A, b
    return out
