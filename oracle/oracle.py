"""ctypes binding of oracle/libkt_oracle.so -- the CPU restatement of the reference (see kt_oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under kintinuous_amd/ may import this module.  Parity status: kernels pinned bit for bit against the
reference's own .cu files built for the host (oracle/ref.py, tests/test_oracle_vs_ref.py); host logic by known-answer tests only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KT_ORACLE_LIB") or os.path.join(_HERE, "libkt_oracle.so")   # KT_ORACLE_LIB: e.g. a sanitizer build


class OIntr(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]

    def level(self, l: int) -> "OIntr":
        d = np.float32(1 << l)
        return OIntr(np.float32(self.fx) / d, np.float32(self.fy) / d, np.float32(self.cx) / d, np.float32(self.cy) / d)


class OMat33(C.Structure):
    _fields_ = [("m", C.c_float * 9)]

    @staticmethod
    def from_np(a) -> "OMat33":
        a = np.asarray(a, dtype=np.float32).reshape(9)
        return OMat33((C.c_float * 9)(*a.tolist()))


class OTrackerConfig(C.Structure):
    _fields_ = [
        ("cols", C.c_int), ("rows", C.c_int), ("N", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("volume_size", C.c_float), ("voxel_shift", C.c_int), ("overlap", C.c_int), ("static_mode", C.c_int),
        ("use_rgbd", C.c_int), ("use_rgbd_icp", C.c_int), ("fast_odometry", C.c_int), ("disable_color_angle", C.c_int),
        ("reduce_order", C.c_int), ("dynamic_cube", C.c_int), ("place_recognition", C.c_int),
    ]


DATATERM_DTYPE = np.dtype([("zero", np.int16, 2), ("one", np.int16, 2), ("diff", np.float32), ("valid", np.uint8), ("pad", np.uint8, 3)])
POINT_DTYPE = np.dtype([("xyz", np.float32, 3), ("pad0", np.float32), ("bgra", np.uint8, 4), ("pad1", np.uint32, 3)])

_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("kt_oracle_kernels.c", "kt_oracle_host.c", "kt_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkt_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.kto_expf.restype = C.c_float
        _lib.kto_expf.argtypes = [C.c_float]
        for n in ("kto_f2i_rn", "kto_f2i_rz", "kto_f2i_rd"):
            getattr(_lib, n).restype = C.c_int
            getattr(_lib, n).argtypes = [C.c_float]
        _lib.kto_integrate_tsdf.restype = C.c_longlong
        _lib.kto_raycast.restype = C.c_longlong
        _lib.kto_extract_cloud_slice.restype = C.c_size_t
        _lib.kto_tracker_create.restype = C.c_void_p
        _lib.kto_tracker_slice_size.restype = C.c_size_t
        _lib.kto_tracker_slice_points.restype = C.c_void_p
        _lib.kto_tracker_volume.restype = C.c_void_p
        _lib.kto_tracker_color_volume.restype = C.c_void_p
        _lib.kto_tracker_vmap_g_prev.restype = C.c_void_p
        _lib.kto_tracker_nmap_g_prev.restype = C.c_void_p
        _lib.kto_tracker_trunc_dist.restype = C.c_float
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f3(a):
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    return (C.c_float * a.size)(*a.tolist())


def _i3(a):
    a = np.asarray(a, dtype=np.int32).reshape(-1)
    return (C.c_int * a.size)(*a.tolist())


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


# ---- image-side ------------------------------------------------------------------------------------
def bilateral_filter(src: np.ndarray) -> np.ndarray:
    src = _c(src, np.uint16)
    rows, cols = src.shape
    dst = np.empty_like(src)
    lib().kto_bilateral_filter(_p(src), _p(dst), cols, rows)
    return dst


def pyr_down(src: np.ndarray) -> np.ndarray:
    src = _c(src, np.uint16)
    rows, cols = src.shape
    dst = np.empty((rows // 2, cols // 2), np.uint16)
    lib().kto_pyr_down(_p(src), cols, rows, _p(dst))
    return dst


def create_vmap(intr: OIntr, depth: np.ndarray, out: np.ndarray = None) -> np.ndarray:
    depth = _c(depth, np.uint16)
    rows, cols = depth.shape
    vmap = np.zeros((3 * rows, cols), np.float32) if out is None else out
    lib().kto_create_vmap(intr, _p(depth), cols, rows, _p(vmap))
    return vmap


def create_nmap(vmap: np.ndarray, out: np.ndarray = None) -> np.ndarray:
    vmap = _c(vmap, np.float32)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    nmap = np.zeros_like(vmap) if out is None else out
    lib().kto_create_nmap(_p(vmap), cols, rows, _p(nmap))
    return nmap


def transform_maps(vmap, nmap, R, t, vout=None, nout=None):
    vmap, nmap = _c(vmap, np.float32), _c(nmap, np.float32)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    vd = np.zeros_like(vmap) if vout is None else vout
    nd = np.zeros_like(nmap) if nout is None else nout
    lib().kto_transform_maps(_p(vmap), _p(nmap), cols, rows, C.byref(OMat33.from_np(R)), _f3(t), _p(vd), _p(nd))
    return vd, nd


def resize_map(inp: np.ndarray, normalize: bool, out: np.ndarray = None) -> np.ndarray:
    inp = _c(inp, np.float32)
    rows, cols = inp.shape[0] // 3, inp.shape[1]
    o = np.zeros((3 * (rows // 2), cols // 2), np.float32) if out is None else out
    (lib().kto_resize_nmap if normalize else lib().kto_resize_vmap)(_p(inp), cols, rows, _p(o))
    return o


def generate_image(vmap, nmap, vmap_color, light_pos, light_number: int = 1):
    vmap, nmap, vmap_color = _c(vmap, np.float32), _c(nmap, np.float32), _c(vmap_color, np.uint8)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    dst, dst_color = np.zeros((rows, cols, 3), np.uint8), np.zeros((rows, cols, 3), np.uint8)
    lib().kto_generate_image(_p(vmap), _p(nmap), _p(vmap_color), cols, rows, _f3(light_pos), light_number, _p(dst), _p(dst_color))
    return dst, dst_color


def generate_depth(R_inv, t, vmap, nmap) -> np.ndarray:
    vmap, nmap = _c(vmap, np.float32), _c(nmap, np.float32)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    dst = np.zeros((rows, cols), np.uint16)
    lib().kto_generate_depth(C.byref(OMat33.from_np(R_inv)), _f3(t), _p(vmap), _p(nmap), cols, rows, _p(dst))
    return dst


def depth_to_metres(src, cutoff: int) -> np.ndarray:
    src = _c(src, np.uint16)
    rows, cols = src.shape
    dst = np.empty((rows, cols), np.float32)
    lib().kto_depth_to_metres(_p(src), _p(dst), cols, rows, cutoff)
    return dst


def bgr_to_intensity(rgb) -> np.ndarray:
    rgb = _c(rgb, np.uint8)
    rows, cols = rgb.shape[:2]
    dst = np.empty((rows, cols), np.uint8)
    lib().kto_bgr_to_intensity(_p(rgb), _p(dst), cols, rows)
    return dst


def pyr_down_gauss_f32(src) -> np.ndarray:
    src = _c(src, np.float32)
    rows, cols = src.shape
    dst = np.empty((rows // 2, cols // 2), np.float32)
    lib().kto_pyr_down_gauss_f32(_p(src), cols, rows, _p(dst))
    return dst


def pyr_down_gauss_u8(src) -> np.ndarray:
    src = _c(src, np.uint8)
    rows, cols = src.shape
    dst = np.empty((rows // 2, cols // 2), np.uint8)
    lib().kto_pyr_down_gauss_u8(_p(src), cols, rows, _p(dst))
    return dst


def derivative_images(src) -> Tuple[np.ndarray, np.ndarray]:
    src = _c(src, np.uint8)
    rows, cols = src.shape
    dx, dy = np.empty((rows, cols), np.int16), np.empty((rows, cols), np.int16)
    lib().kto_derivative_images(_p(src), cols, rows, _p(dx), _p(dy))
    return dx, dy


def project_to_cloud(depth, fx, fy, cx, cy, level) -> np.ndarray:
    depth = _c(depth, np.float32)
    rows, cols = depth.shape
    cloud = np.empty((rows, cols, 3), np.float32)
    lib().kto_project_to_cloud(_p(depth), cols, rows, _p(cloud), C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), level)
    return cloud


# ---- tracking --------------------------------------------------------------------------------------
def icp_step(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr: OIntr, vmap_g_prev, nmap_g_prev, dist_thres, angle_thres,
             order: int = 0):
    vmap_curr, nmap_curr = _c(vmap_curr, np.float32), _c(nmap_curr, np.float32)
    vmap_g_prev, nmap_g_prev = _c(vmap_g_prev, np.float32), _c(nmap_g_prev, np.float32)
    rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
    A, b, r = (C.c_float * 36)(), (C.c_float * 6)(), (C.c_float * 2)()
    lib().kto_icp_step(C.byref(OMat33.from_np(Rcurr)), _f3(tcurr), _p(vmap_curr), _p(nmap_curr), C.byref(OMat33.from_np(Rprev_inv)),
                       _f3(tprev), intr, _p(vmap_g_prev), _p(nmap_g_prev), cols, rows, C.c_float(dist_thres), C.c_float(angle_thres),
                       order, A, b, r)
    return np.array(A, np.float32).reshape(6, 6), np.array(b, np.float32), np.array(r, np.float32)


def rgb_residual(min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, max_depth_delta, kt, krkinv):
    dIdx, dIdy = _c(dIdx, np.int16), _c(dIdy, np.int16)
    last_depth, next_depth = _c(last_depth, np.float32), _c(next_depth, np.float32)
    last_image, next_image = _c(last_image, np.uint8), _c(next_image, np.uint8)
    rows, cols = next_image.shape
    corres = np.zeros((rows, cols), DATATERM_DTYPE)
    sigma, count = C.c_int(0), C.c_int(0)
    lib().kto_rgb_residual(C.c_float(min_scale), _p(dIdx), _p(dIdy), _p(last_depth), _p(next_depth), _p(last_image), _p(next_image),
                           cols, rows, _p(corres), C.c_float(max_depth_delta), _f3(kt), C.byref(OMat33.from_np(krkinv)),
                           C.byref(sigma), C.byref(count))
    return corres, sigma.value, count.value


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale, order: int = 0):
    corres = np.ascontiguousarray(corres)
    cloud, dIdx, dIdy = _c(cloud, np.float32), _c(dIdx, np.int16), _c(dIdy, np.int16)
    rows, cols = dIdx.shape
    A, b = (C.c_float * 36)(), (C.c_float * 6)()
    lib().kto_rgb_step(_p(corres), C.c_float(sigma), _p(cloud), C.c_float(fx), C.c_float(fy), _p(dIdx), _p(dIdy), C.c_float(sobel_scale),
                       cols, rows, order, A, b)
    return np.array(A, np.float32).reshape(6, 6), np.array(b, np.float32)


# ---- volume ----------------------------------------------------------------------------------------
def scale_depth(depth, intr: OIntr, angle_color: bool) -> np.ndarray:
    depth = _c(depth, np.uint16)
    rows, cols = depth.shape
    out = np.empty((rows, cols), np.float32)
    lib().kto_scale_depth(_p(depth), _p(out), cols, rows, intr, int(angle_color))
    return out


def integrate_tsdf(depth, intr: OIntr, volume_size, Rcurr_inv, tcurr, tranc_dist, volume, voxel_wrap, color_volume, colors, nmap_curr,
                   angle_color: bool) -> Tuple[int, np.ndarray]:
    """In-place on volume (int16 [N,N,N]) and color_volume (uint8 [N,N,N,4]); returns (U, depth_scaled)."""
    depth, colors, nmap_curr = _c(depth, np.uint16), _c(colors, np.uint8), _c(nmap_curr, np.float32)
    assert volume.flags.c_contiguous and color_volume.flags.c_contiguous
    rows, cols = depth.shape
    N = volume.shape[0]
    scaled = np.empty((rows, cols), np.float32)
    U = lib().kto_integrate_tsdf(_p(depth), cols, rows, intr, _f3(volume_size), C.byref(OMat33.from_np(Rcurr_inv)), _f3(tcurr),
                                 C.c_float(tranc_dist), _p(volume), _p(scaled), _i3(voxel_wrap), _p(color_volume), _p(colors),
                                 _p(nmap_curr), int(angle_color), N)
    return int(U), scaled


def raycast(intr: OIntr, Rcurr, tcurr, tranc_dist, volume_size, volume, vmap, nmap, voxel_wrap, vmap_color, color_volume) -> int:
    """In-place on vmap, nmap ([3*rows, cols] float32) and vmap_color ([rows, cols, 4] uint8); returns S."""
    assert vmap.flags.c_contiguous and nmap.flags.c_contiguous and vmap_color.flags.c_contiguous
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    N = volume.shape[0]
    S = lib().kto_raycast(intr, C.byref(OMat33.from_np(Rcurr)), _f3(tcurr), C.c_float(tranc_dist), _f3(volume_size), _p(volume), _p(vmap),
                          _p(nmap), cols, rows, _i3(voxel_wrap), _p(vmap_color), _p(color_volume), N)
    return int(S)


def clear_volume(vol: np.ndarray, axis: int, back: bool, current_wrap: int, delta_wrap: int) -> None:
    N = vol.shape[0]
    elem = 2 if vol.dtype == np.int16 else 4
    lib().kto_clear_volume(_p(vol), elem, N, axis, int(back), current_wrap, delta_wrap)


def extract_cloud_slice(volume, volume_size, cap, voxel_wrap, color_volume, minX, maxX, minY, maxY, minZ, maxZ, subsample, real_wrap):
    N = volume.shape[0]
    out = np.zeros(cap, POINT_DTYPE)
    n = lib().kto_extract_cloud_slice(_p(volume), _f3(volume_size), _p(out), C.c_size_t(cap), _i3(voxel_wrap), _p(color_volume), minX, maxX,
                                      minY, maxY, minZ, maxZ, subsample, _i3(real_wrap), N)
    return out[: int(n)]


NPOINT_DTYPE = np.dtype([("xyz", np.float32, 3), ("one", np.float32), ("normal", np.float32, 3), ("zero", np.float32), ("bgra", np.uint8, 4),
                         ("curvature", np.float32), ("pad", np.float32, 2)])


def slice_process(points, weight_cull: int, leaf: float, k: int = 20) -> np.ndarray:
    """CloudSliceProcessor's per-slice stage (weight cull, VoxelGrid at `leaf`, kNN(k) normals) -> pcl::PointXYZRGBNormal records."""
    points = np.ascontiguousarray(points)
    assert points.dtype == POINT_DTYPE
    out = np.zeros(max(len(points), 1), NPOINT_DTYPE)
    lib().kto_slice_process.restype = C.c_size_t
    n = lib().kto_slice_process(_p(points), C.c_size_t(len(points)), int(weight_cull), C.c_float(leaf), int(k), _p(out))
    return out[: int(n)]


def voxel_grid_normal(points, leaf: float) -> np.ndarray:
    """CloudSliceProcessor::save's final pcl::VoxelGrid<pcl::PointXYZRGBNormal> (every field averaged, leaves in key order)."""
    points = np.ascontiguousarray(points)
    assert points.dtype == NPOINT_DTYPE
    out = np.zeros(max(len(points), 1), NPOINT_DTYPE)
    lib().kto_voxel_grid_normal.restype = C.c_size_t
    n = lib().kto_voxel_grid_normal(_p(points), C.c_size_t(len(points)), C.c_float(leaf), _p(out))
    return out[: int(n)]


def pcd_binary(points) -> bytes:
    """pcl::io::savePCDFile(file, cloud, true) for pcl::PointXYZRGBNormal: header + 32 bytes per point."""
    points = np.ascontiguousarray(points)
    assert points.dtype == NPOINT_DTYPE
    out = np.zeros(512 + 32 * len(points), np.uint8)
    lib().kto_pcd_binary.restype = C.c_size_t
    n = lib().kto_pcd_binary(_p(points), C.c_size_t(len(points)), _p(out))
    return out[: int(n)].tobytes()


# ---- host math -------------------------------------------------------------------------------------
def mat33_inverse(R) -> np.ndarray:
    o = OMat33()
    lib().kto_mat33_inverse(C.byref(OMat33.from_np(R)), C.byref(o))
    return np.array(o.m, np.float32).reshape(3, 3)


def ldlt_solve6(A, b) -> np.ndarray:
    A = np.ascontiguousarray(A, np.float64).reshape(36)
    b = np.ascontiguousarray(b, np.float64).reshape(6)
    x = np.zeros(6, np.float64)
    lib().kto_ldlt_solve6(_p(A), _p(b), _p(x))
    return x


def rodrigues(r) -> np.ndarray:
    r = np.ascontiguousarray(r, np.float64).reshape(3)
    R = np.zeros(9, np.float64)
    lib().kto_rodrigues(_p(r), _p(R))
    return R.reshape(3, 3)


def reposition_cube(R, tlast, volume_size, voxel_size, thresh, basis) -> np.ndarray:
    """KintinuousTracker::repositionCube on explicit state; returns the (possibly moved) basis."""
    b = (C.c_float * 3)(*[float(v) for v in basis])
    lib().kto_reposition_cube(_f3(np.asarray(R, np.float32).reshape(-1)), _f3(tlast), C.c_float(volume_size), _f3(voxel_size), int(thresh), b)
    return np.array(b, np.float32)


def quat_from_mat33(R) -> np.ndarray:
    q = (C.c_float * 4)()
    lib().kto_quat_from_mat33(C.byref(OMat33.from_np(R)), q)
    return np.array(q, np.float32)


# ---- tracker ---------------------------------------------------------------------------------------
class OracleTracker:
    def __init__(self, cfg: OTrackerConfig):
        self.cfg = cfg
        self.h = C.c_void_p(lib().kto_tracker_create(C.byref(cfg)))

    def close(self):
        if self.h:
            lib().kto_tracker_destroy(self.h)
            self.h = None

    def process_frame(self, depth, rgb, ts: int) -> None:
        depth, rgb = _c(depth, np.uint16), _c(rgb, np.uint8)
        lib().kto_tracker_process_frame(self.h, _p(depth), _p(rgb), C.c_uint64(ts))

    def load_trajectory(self, utimes, pose7) -> None:
        """-p ground truth: utimes[n] (uint64), pose7[n,7] = x y z qx qy qz qw (float32)."""
        utimes, pose7 = _c(utimes, np.uint64), _c(pose7, np.float32)
        lib().kto_tracker_load_trajectory(self.h, C.c_int(len(utimes)), _p(utimes), _p(pose7))

    def finalise(self) -> None:
        lib().kto_tracker_finalise(self.h)

    def volume_basis(self) -> np.ndarray:
        b = (C.c_float * 3)()
        lib().kto_tracker_get_volume_basis(self.h, b)
        return np.array(b, np.float32)

    def pose(self):
        R, t, g = (C.c_float * 9)(), (C.c_float * 3)(), (C.c_float * 3)()
        lib().kto_tracker_get_pose(self.h, R, t, g)
        return np.array(R, np.float32).reshape(3, 3), np.array(t, np.float32), np.array(g, np.float32)

    def num_poses(self) -> int:
        return lib().kto_tracker_num_poses(self.h)

    def dense_pose(self, i):
        ts, p, il = C.c_uint64(0), (C.c_float * 16)(), C.c_int(0)
        lib().kto_tracker_get_dense_pose(self.h, i, C.byref(ts), p, C.byref(il))
        return ts.value, np.array(p, np.float32).reshape(4, 4), bool(il.value)

    def voxel_wrap(self):
        w = (C.c_int * 3)()
        lib().kto_tracker_get_voxel_wrap(self.h, w)
        return np.array(w, np.int32)

    def num_slices(self) -> int:
        return lib().kto_tracker_num_slices(self.h)

    def slice(self, i):
        n = lib().kto_tracker_slice_size(self.h, i)
        dim = lib().kto_tracker_slice_dimension(self.h, i)
        ptr = lib().kto_tracker_slice_points(self.h, i)
        if n == 0:
            return np.zeros(0, POINT_DTYPE), dim
        buf = (C.c_char * (n * 32)).from_address(ptr)
        return np.frombuffer(buf, dtype=POINT_DTYPE, count=n).copy(), dim

    def pr_samples(self):
        """[(utime, trans[3], rot[3,3])] of the frames sampled for place recognition, in order."""
        out = []
        for i in range(lib().kto_tracker_num_pr_samples(self.h)):
            ut, tr, ro = C.c_uint64(0), (C.c_float * 3)(), (C.c_float * 9)()
            lib().kto_tracker_pr_sample(self.h, i, C.byref(ut), tr, ro)
            out.append((ut.value, np.array(tr, np.float32), np.array(ro, np.float32).reshape(3, 3)))
        return out

    def slice_pr_id(self, i) -> int:
        return lib().kto_tracker_slice_pr_id(self.h, i)

    def _arr(self, ptr, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = (C.c_char * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def volume(self):
        N = self.cfg.N
        return self._arr(lib().kto_tracker_volume(self.h), np.int16, (N, N, N))

    def color_volume(self):
        N = self.cfg.N
        return self._arr(lib().kto_tracker_color_volume(self.h), np.uint8, (N, N, N, 4))

    def vmap_g_prev(self, level=0):
        c, r = self.cfg.cols >> level, self.cfg.rows >> level
        return self._arr(lib().kto_tracker_vmap_g_prev(self.h, level), np.float32, (3 * r, c))

    def nmap_g_prev(self, level=0):
        c, r = self.cfg.cols >> level, self.cfg.rows >> level
        return self._arr(lib().kto_tracker_nmap_g_prev(self.h, level), np.float32, (3 * r, c))

    def trunc_dist(self) -> float:
        return float(lib().kto_tracker_trunc_dist(self.h))

    def stage_seconds(self):
        s = (C.c_double * 6)()
        lib().kto_tracker_stage_seconds(self.h, s)
        return dict(zip(("pyramid", "odometry", "shift", "integrate", "raycast", "resize"), [float(v) for v in s]))

    def last_counts(self):
        U, S = C.c_longlong(0), C.c_longlong(0)
        lib().kto_tracker_last_counts(self.h, C.byref(U), C.byref(S))
        return int(U.value), int(S.value)
