"""ctypes binding of libkt_hip.so (include/kt_abi.h) -- used by tests/, bench.py and the smoke test.

This is plumbing only: device buffers are raw HIP allocations made through the C-ABI (kt_malloc) and
moved with kt_upload / kt_download; numpy arrays live on the host.  There is NO CPU fallback: if the HIP
library is missing or fails to load, importing `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KT_HIP_LIB") or os.path.join(_HERE, "libkt_hip.so")   # KT_HIP_LIB: an alternative build (A/B experiments)

KT_OK = 0


class KtError(RuntimeError):
    pass


class Intr(C.Structure):  # kt_intr
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]

    def level(self, l: int) -> "Intr":  # Intr::operator(), internal.h:255-259 (float division)
        d = np.float32(1 << l)
        return Intr(np.float32(self.fx) / d, np.float32(self.fy) / d, np.float32(self.cx) / d, np.float32(self.cy) / d)


class Mat33(C.Structure):  # kt_mat33
    _fields_ = [("m", C.c_float * 9)]

    @staticmethod
    def from_np(a) -> "Mat33":
        a = np.asarray(a, dtype=np.float32).reshape(9)
        return Mat33((C.c_float * 9)(*a.tolist()))


class TrackerConfig(C.Structure):  # kt_tracker_config
    _fields_ = [
        ("cols", C.c_int), ("rows", C.c_int), ("N", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("volume_size", C.c_float), ("voxel_shift", C.c_int), ("overlap", C.c_int), ("static_mode", C.c_int),
        ("use_rgbd", C.c_int), ("use_rgbd_icp", C.c_int), ("fast_odometry", C.c_int), ("disable_color_angle", C.c_int),
        ("max_slice_points", C.c_int), ("dynamic_cube", C.c_int), ("place_recognition", C.c_int),
    ]


DATATERM_DTYPE = np.dtype([("zero", np.int16, 2), ("one", np.int16, 2), ("diff", np.float32), ("valid", np.uint8), ("pad", np.uint8, 3)])
POINT_DTYPE = np.dtype([("xyz", np.float32, 3), ("pad0", np.float32), ("bgra", np.uint8, 4), ("pad1", np.uint32, 3)])
assert DATATERM_DTYPE.itemsize == 16 and POINT_DTYPE.itemsize == 32

_vp, _i, _f, _d, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint64
_pf = C.POINTER(C.c_float)
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int)
_pI = C.POINTER(Intr)
_pM = C.POINTER(Mat33)

_PROTOS = {
    "kt_last_error": (C.c_char_p, []),
    "kt_version": (C.c_char_p, []),
    "kt_device_count": (_i, [_pi]),
    "kt_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "kt_ctx_destroy": (_i, [_vp]),
    "kt_ctx_set_stream": (_i, [_vp, _vp]),
    "kt_ctx_stream": (_vp, [_vp]),
    "kt_sync": (_i, [_vp]),
    "kt_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "kt_free": (_i, [_vp, _vp]),
    "kt_memset": (_i, [_vp, _vp, _i, _sz]),
    "kt_upload": (_i, [_vp, _vp, _vp, _sz]),
    "kt_download": (_i, [_vp, _vp, _vp, _sz]),
    "kt_upload2d": (_i, [_vp, _vp, _vp, _sz, _sz, _i]),
    "kt_download2d": (_i, [_vp, _vp, _sz, _vp, _sz, _i]),
    "kt_bilateral_filter": (_i, [_vp, _vp, _vp, _i, _i]),
    "kt_pyr_down": (_i, [_vp, _vp, _i, _i, _vp]),
    "kt_create_vmap": (_i, [_vp, _pI, _vp, _i, _i, _vp]),
    "kt_create_nmap": (_i, [_vp, _vp, _i, _i, _vp]),
    "kt_transform_maps": (_i, [_vp, _vp, _vp, _i, _i, _pM, _pf, _vp, _vp]),
    "kt_resize_vmap": (_i, [_vp, _vp, _i, _i, _vp]),
    "kt_resize_nmap": (_i, [_vp, _vp, _i, _i, _vp]),
    "kt_generate_image": (_i, [_vp, _vp, _vp, _vp, _i, _i, _pf, _i, _vp, _vp]),
    "kt_generate_depth": (_i, [_vp, _pM, _pf, _vp, _vp, _i, _i, _vp]),
    "kt_depth_to_metres": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "kt_bgr_to_intensity": (_i, [_vp, _vp, _vp, _i, _i]),
    "kt_pyr_down_gauss_f32": (_i, [_vp, _vp, _i, _i, _vp]),
    "kt_pyr_down_gauss_u8": (_i, [_vp, _vp, _i, _i, _vp]),
    "kt_derivative_images": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "kt_project_to_cloud": (_i, [_vp, _vp, _i, _i, _vp, _d, _d, _d, _d, _i]),
    "kt_build_pyramid": (_i, [_vp, _pI, _vp, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "kt_icp_step": (_i, [_vp, _pM, _pf, _vp, _vp, _pM, _pf, _pI, _vp, _vp, _i, _i, _f, _f, _pf, _pf, _pf]),
    "kt_rgb_residual": (_i, [_vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _f, _pf, _pM, _pi, _pi]),
    "kt_rgb_step": (_i, [_vp, _vp, _f, _vp, _f, _f, _vp, _vp, _f, _i, _i, _pf, _pf]),
    "kt_init_volume": (_i, [_vp, _vp, _i]),
    "kt_init_color_volume": (_i, [_vp, _vp, _i]),
    "kt_icp_track": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _pI, _pM, _pf, _pi, _f, _f, _pM, _pf, _pf, _pf]),
    "kt_integrate_tsdf": (_i, [_vp, _vp, _i, _i, _pI, _pf, _pM, _pf, _f, _vp, _vp, _pi, _vp, _vp, _vp, _i, _i]),
    "kt_raycast": (_i, [_vp, _pI, _pM, _pf, _f, _pf, _vp, _vp, _vp, _i, _i, _pi, _vp, _vp, _i]),
    "kt_clear_volume": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i]),
    "kt_extract_cloud_slice": (_i, [_vp, _vp, _pf, _vp, _sz, _pi, _vp, _i, _i, _i, _i, _i, _i, _i, _pi, _i, C.POINTER(_sz)]),
    "kt_tracker_create": (_i, [_vp, C.POINTER(TrackerConfig), C.POINTER(_vp)]),
    "kt_tracker_destroy": (_i, [_vp]),
    "kt_tracker_reset": (_i, [_vp]),
    "kt_tracker_process_frame": (_i, [_vp, _vp, _vp, _u64]),
    "kt_tracker_process_frame_host": (_i, [_vp, _vp, _vp, _u64]),
    "kt_tracker_load_trajectory": (_i, [_vp, _i, _vp, _vp]),
    "kt_tracker_finalise": (_i, [_vp]),
    "kt_tracker_get_volume_basis": (_i, [_vp, _pf]),
    "kt_host_reposition_cube": (None, [_pf, _pf, _f, _pf, _i, _pf]),
    "kt_tracker_get_pose": (_i, [_vp, _pf, _pf, _pf]),
    "kt_tracker_num_poses": (_i, [_vp]),
    "kt_tracker_get_dense_pose": (_i, [_vp, _i, C.POINTER(_u64), _pf, _pi]),
    "kt_tracker_get_voxel_wrap": (_i, [_vp, _pi]),
    "kt_tracker_num_slices": (_i, [_vp]),
    "kt_tracker_slice_info": (_i, [_vp, _i, C.POINTER(_sz), _pi]),
    "kt_tracker_slice_points": (_i, [_vp, _i, _vp]),
    "kt_tracker_volume": (_vp, [_vp]),
    "kt_tracker_color_volume": (_vp, [_vp]),
    "kt_tracker_vmap_g_prev": (_vp, [_vp, _i]),
    "kt_tracker_vmap_curr_color": (_vp, [_vp]),
    "kt_tracker_nmap_g_prev": (_vp, [_vp, _i]),
    "kt_tracker_trunc_dist": (_f, [_vp]),
    "kt_tracker_enable_profiling": (_i, [_vp, _i]),
    "kt_tracker_stage_ms": (_i, [_vp, _pf]),
    "kt_tracker_stage_counts": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "kt_tracker_enable_counts": (_i, [_vp, _i]),
    "kt_tracker_last_counts": (_i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "kt_tracker_debug_counts": (_i, [_vp, C.POINTER(C.c_uint)]),
    "kt_tracker_debug_state": (_i, [_vp, _pf]),
    "kt_tracker_plan_stats": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "kt_tracker_odometry_fallbacks": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "kt_debug_wait_limit": (_i, [_vp, C.c_uint]),
    "kt_tracker_debug_side_gate": (_i, [_vp]),
    "kt_tracker_enable_slice_stage": (_i, [_vp, _i, _i, _i]),
    "kt_tracker_slice_processed_info": (_i, [_vp, _i, C.POINTER(C.c_longlong)]),
    "kt_tracker_slice_processed": (_i, [_vp, _i, _vp]),
    "kt_tracker_debug_pose_log": (_i, [_vp, _i, _pf, _i, C.POINTER(C.c_int)]),
    "kt_tracker_debug_plan_truth": (_i, [_vp, _pf, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint]),
    "kt_debug_tsdf_lean": (_i, [_i]),
    "kt_debug_tsdf_kernel": (C.c_char_p, []),
    "kt_debug_tsdf_contract": (_i, [_i]),
    "kt_debug_sq_threshold": (_f, [_f, _i]),
    "kt_debug_icp_levels": (_i, [_i]),
    "kt_debug_ri_levels": (_i, [_i]),
    "kt_tracker_debug_icp_levels": (_i, [_vp]),
    "kt_debug_icp_wg_times": (_i, [_vp, C.POINTER(C.c_ulonglong)]),
    "kt_debug_tsdf_timeline": (_i, [_vp, C.POINTER(C.c_ulonglong), _i]),
    "kt_debug_solve_check": (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(_i)]),
    "kt_debug_unpack_table": (_i, [_vp, _pf]),
    "kt_debug_rcp_check": (_i, [_vp, C.POINTER(C.c_uint)]),
    "kt_debug_handoff_fault": (_i, [_vp, _i, _i, C.c_uint, C.POINTER(C.c_uint)]),
    "kt_tracker_num_pr_samples": (_i, [_vp]),
    "kt_tracker_pr_sample": (_i, [_vp, _i, C.POINTER(_u64), _pf, _pf, C.POINTER(C.c_int)]),
    "kt_tracker_slice_pr_id": (_i, [_vp, _i, C.POINTER(C.c_int)]),
    "kt_host_place_recognition_movement": (_f, [_pf, _pf, _pf, _pf]),
    "kt_slice_process": (_i, [_vp, _vp, _sz, _i, _f, _i, _vp, C.POINTER(_sz)]),
    "kt_slice_ws_create": (_i, [_vp, _sz, _vp, C.POINTER(_vp)]),
    "kt_slice_ws_destroy": (_i, [_vp]),
    "kt_slice_ws_stream": (_vp, [_vp]),
    "kt_slice_process_device": (_i, [_vp, _vp, _vp, _sz, _i, _f, _i]),
    "kt_slice_ws_count": (_i, [_vp, C.POINTER(_sz)]),
    "kt_slice_ws_output": (_vp, [_vp]),
    "kt_host_voxel_grid_normal": (_i, [_vp, _sz, _f, _vp, C.POINTER(_sz)]),
    "kt_host_save_pcd": (_i, [C.c_char_p, _vp, _sz]),
    "kt_comm_unique_id": (_i, [C.POINTER(C.c_ubyte)]),
    "kt_comm_init": (_i, [_vp, _i, _i, C.POINTER(C.c_ubyte), C.POINTER(_vp)]),
    "kt_pose_gather": (_i, [_vp, _vp, _i, _pf]),
    "kt_comm_barrier": (_i, [_vp]),
    "kt_comm_destroy": (_i, [_vp]),
    "kt_tracker_host_times": (_i, [_vp, _pd, _i]),
    "kt_tracker_prefetch_frame": (_i, [_vp, _vp, _vp]),
    "kt_tracker_prefetch_frame_host": (_i, [_vp, _vp, _vp]),
    "kt_tracker_slice_pose": (_i, [_vp, _i, _pf, _pf, C.POINTER(_u64)]),
    "kt_tracker_set_parked": (_i, [_vp, _i]),
    "kt_host_ldlt_solve6": (_i, [_pd, _pd, _pd]),
    "kt_host_rodrigues": (_i, [_pd, _pd]),
    "kt_host_mat33_inverse": (_i, [_pf, _pf]),
    "kt_host_pose_update": (_i, [_pd, _pd, _pf, _pf, _pf, _pf]),
    "kt_host_compute_krk": (_i, [_pd, C.c_double, C.c_double, C.c_double, C.c_double, _pf, _pf]),
    "kt_host_trajectory_pose": (None, [_pf, _pf]),
    "kt_host_ground_truth_pose": (None, [_pf, _pf, _pf, _pf, _pf, _pf]),
    "kt_tracker_export_poses_device": (_i, [_vp, _i, _vp]),
}

ABI_SYMBOLS = tuple(_PROTOS.keys())

# libkt_debug.so (csrc/kt_measure.h): measurement kernels kept out of the product library
_MEASURE_PROTOS = {
    "kt_debug_stream": (_i, [_vp, _vp, _sz, _i, _i]),
    "kt_debug_stream_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _i]),
    "kt_debug_valu_rates": (_i, [_vp, _i, _i, _i, _pd]),
    "kt_debug_div_check": (_i, [_vp, C.POINTER(C.c_uint)]),
}
MEASURE_SYMBOLS = tuple(_MEASURE_PROTOS.keys())
MEASURE_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkt_debug.so")
_mlib = None


def measure_lib() -> C.CDLL:
    """Load libkt_debug.so (next to libkt_hip.so, which it links against)."""
    global _mlib
    if _mlib is None:
        lib()
        if not os.path.exists(MEASURE_LIB_PATH):
            raise KtError(f"{MEASURE_LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` first")
        l = C.CDLL(MEASURE_LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _MEASURE_PROTOS.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _mlib = l
    return _mlib

_lib = None


def lib() -> C.CDLL:
    """Load libkt_hip.so (built by __graft_entry__.build()).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KtError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` first")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _PROTOS.items():
            if name.startswith("kt_debug_") and os.environ.get("KT_HIP_LIB") and not hasattr(l, name):
                continue   # an A/B build of an older tree may lack a newer test hook (never the tree's own library)
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _chk(status: int) -> None:
    if status != KT_OK:
        raise KtError(f"kt status {status}: {lib().kt_last_error().decode(errors='replace')}")


def _fp(a) -> "C.Array":
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    return (C.c_float * a.size)(*a.tolist())


def _ip(a) -> "C.Array":
    a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
    return (C.c_int * a.size)(*a.tolist())


def _count(n: int) -> int:
    """kt_tracker_num_*: a negative count means the frame in flight could not be completed"""
    if n < 0:
        raise KtError(lib().kt_last_error().decode() or "tracker error")
    return n


class DevBuf:
    """A raw device allocation owned through kt_malloc / kt_free."""

    def __init__(self, ctx: "Ctx", nbytes: int, ptr: Optional[int] = None):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.owned = ptr is None
        if ptr is None:
            p = _vp()
            _chk(lib().kt_malloc(ctx.h, self.nbytes, C.byref(p)))
            ptr = p.value
        self.ptr = ptr

    def free(self) -> None:
        if self.owned and self.ptr:
            lib().kt_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Ctx:
    """kt_ctx: one device + one stream."""

    def __init__(self, device: int = 0):
        h = _vp()
        _chk(lib().kt_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self) -> None:
        if self.h:
            lib().kt_ctx_destroy(self.h)
            self.h = None

    def sync(self) -> None:
        _chk(lib().kt_sync(self.h))

    # ---- memory --------------------------------------------------------------------------------
    def empty(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def zeros(self, nbytes: int) -> DevBuf:
        b = DevBuf(self, nbytes)
        _chk(lib().kt_memset(self.h, b.ptr, 0, nbytes))
        return b

    def upload(self, a: np.ndarray, into: Optional[DevBuf] = None) -> DevBuf:
        a = np.ascontiguousarray(a)
        b = into if into is not None else DevBuf(self, a.nbytes)
        _chk(lib().kt_upload(self.h, b.ptr, a.ctypes.data, a.nbytes))
        return b

    def download(self, b, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        ptr = b.ptr if isinstance(b, DevBuf) else int(b)
        _chk(lib().kt_download(self.h, out.ctypes.data, ptr, out.nbytes))
        return out

    # ---- image ops -----------------------------------------------------------------------------
    def bilateral_filter(self, src: DevBuf, dst: DevBuf, cols: int, rows: int) -> None:
        _chk(lib().kt_bilateral_filter(self.h, src.ptr, dst.ptr, cols, rows))

    def pyr_down(self, src: DevBuf, scols: int, srows: int, dst: DevBuf) -> None:
        _chk(lib().kt_pyr_down(self.h, src.ptr, scols, srows, dst.ptr))

    def create_vmap(self, intr: Intr, depth: DevBuf, cols: int, rows: int, vmap: DevBuf) -> None:
        _chk(lib().kt_create_vmap(self.h, C.byref(intr), depth.ptr, cols, rows, vmap.ptr))

    def create_nmap(self, vmap: DevBuf, cols: int, rows: int, nmap: DevBuf) -> None:
        _chk(lib().kt_create_nmap(self.h, vmap.ptr, cols, rows, nmap.ptr))

    def transform_maps(self, vs, ns, cols, rows, R, t, vd, nd) -> None:
        _chk(lib().kt_transform_maps(self.h, vs.ptr, ns.ptr, cols, rows, C.byref(Mat33.from_np(R)), _fp(t), vd.ptr, nd.ptr))

    def resize_vmap(self, src, in_cols, in_rows, dst) -> None:
        _chk(lib().kt_resize_vmap(self.h, src.ptr, in_cols, in_rows, dst.ptr))

    def resize_nmap(self, src, in_cols, in_rows, dst) -> None:
        _chk(lib().kt_resize_nmap(self.h, src.ptr, in_cols, in_rows, dst.ptr))

    def generate_image(self, vmap, nmap, vmap_color, cols, rows, light_pos, light_number, dst, dst_color) -> None:
        _chk(lib().kt_generate_image(self.h, vmap.ptr, nmap.ptr, vmap_color.ptr, cols, rows, _fp(light_pos), light_number, dst.ptr, dst_color.ptr))

    def generate_depth(self, R_inv, t, vmap, nmap, cols, rows, dst) -> None:
        _chk(lib().kt_generate_depth(self.h, C.byref(Mat33.from_np(R_inv)), _fp(t), vmap.ptr, nmap.ptr, cols, rows, dst.ptr))

    def depth_to_metres(self, src, dst, cols, rows, cutoff) -> None:
        _chk(lib().kt_depth_to_metres(self.h, src.ptr, dst.ptr, cols, rows, cutoff))

    def bgr_to_intensity(self, src, dst, cols, rows) -> None:
        _chk(lib().kt_bgr_to_intensity(self.h, src.ptr, dst.ptr, cols, rows))

    def pyr_down_gauss_f32(self, src, scols, srows, dst) -> None:
        _chk(lib().kt_pyr_down_gauss_f32(self.h, src.ptr, scols, srows, dst.ptr))

    def pyr_down_gauss_u8(self, src, scols, srows, dst) -> None:
        _chk(lib().kt_pyr_down_gauss_u8(self.h, src.ptr, scols, srows, dst.ptr))

    def derivative_images(self, src, cols, rows, dx, dy) -> None:
        _chk(lib().kt_derivative_images(self.h, src.ptr, cols, rows, dx.ptr, dy.ptr))

    def project_to_cloud(self, depth, cols, rows, cloud, fx, fy, cx, cy, level) -> None:
        _chk(lib().kt_project_to_cloud(self.h, depth.ptr, cols, rows, cloud.ptr, fx, fy, cx, cy, level))

    def build_pyramid(self, intr: Intr, depth0, cols, rows, depths_out, vmaps, nmaps) -> None:
        d = (_vp * 3)(*[b.ptr for b in depths_out])
        v = (_vp * 4)(*[b.ptr for b in vmaps])
        n = (_vp * 4)(*[b.ptr for b in nmaps])
        _chk(lib().kt_build_pyramid(self.h, C.byref(intr), depth0.ptr, cols, rows, d, v, n))

    # ---- tracking ------------------------------------------------------------------------------
    def icp_step(self, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr: Intr, vmap_g_prev, nmap_g_prev, cols, rows,
                 dist_thres, angle_thres) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        A = (C.c_float * 36)()
        b = (C.c_float * 6)()
        r = (C.c_float * 2)()
        _chk(lib().kt_icp_step(self.h, C.byref(Mat33.from_np(Rcurr)), _fp(tcurr), vmap_curr.ptr, nmap_curr.ptr,
                               C.byref(Mat33.from_np(Rprev_inv)), _fp(tprev), C.byref(intr), vmap_g_prev.ptr, nmap_g_prev.ptr,
                               cols, rows, dist_thres, angle_thres, A, b, r))
        return (np.array(A, dtype=np.float32).reshape(6, 6), np.array(b, dtype=np.float32), np.array(r, dtype=np.float32))

    def icp_track(self, vmaps_curr, nmaps_curr, vmaps_g_prev, nmaps_g_prev, cols, rows, intr: Intr, Rprev, tprev, iterations,
                  dist_thres, angle_thres):
        """kt_icp_track: ICPOdometry::getIncrementalTransformation in one call.  maps: lists of 4 DevBuf (None where iterations[l] == 0).
        Returns (Rcurr, tcurr, A_last[6, 6], residual[2])."""
        def arr(bufs):
            return (_vp * 4)(*[(b.ptr if b is not None else None) for b in bufs])
        Rc = Mat33()
        tc, A, r = (C.c_float * 3)(), (C.c_float * 36)(), (C.c_float * 2)()
        it = (C.c_int * 4)(*[int(v) for v in iterations])
        _chk(lib().kt_icp_track(self.h, arr(vmaps_curr), arr(nmaps_curr), arr(vmaps_g_prev), arr(nmaps_g_prev), cols, rows, C.byref(intr),
                                C.byref(Mat33.from_np(Rprev)), _fp(tprev), it, dist_thres, angle_thres, C.byref(Rc), tc, A, r))
        return (np.array(Rc.m, np.float32).reshape(3, 3), np.array(tc, np.float32), np.array(A, np.float32).reshape(6, 6), np.array(r, np.float32))

    def rgb_residual(self, min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, cols, rows, corres,
                     max_depth_delta, kt, krkinv) -> Tuple[int, int]:
        sigma, count = C.c_int(0), C.c_int(0)
        _chk(lib().kt_rgb_residual(self.h, min_scale, dIdx.ptr, dIdy.ptr, last_depth.ptr, next_depth.ptr, last_image.ptr,
                                   next_image.ptr, cols, rows, corres.ptr, max_depth_delta, _fp(kt),
                                   C.byref(Mat33.from_np(krkinv)), C.byref(sigma), C.byref(count)))
        return sigma.value, count.value

    def rgb_step(self, corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale, cols, rows) -> Tuple[np.ndarray, np.ndarray]:
        A = (C.c_float * 36)()
        b = (C.c_float * 6)()
        _chk(lib().kt_rgb_step(self.h, corres.ptr, sigma, cloud.ptr, fx, fy, dIdx.ptr, dIdy.ptr, sobel_scale, cols, rows, A, b))
        return np.array(A, dtype=np.float32).reshape(6, 6), np.array(b, dtype=np.float32)

    # ---- volume --------------------------------------------------------------------------------
    def init_volume(self, vol: DevBuf, N: int) -> None:
        _chk(lib().kt_init_volume(self.h, vol.ptr, N))

    def init_color_volume(self, cvol: DevBuf, N: int) -> None:
        _chk(lib().kt_init_color_volume(self.h, cvol.ptr, N))

    def integrate_tsdf(self, depth, cols, rows, intr: Intr, volume_size, Rcurr_inv, tcurr, tranc_dist, volume, depth_scaled,
                       voxel_wrap, color_volume, colors, nmap_curr, angle_color, N) -> None:
        _chk(lib().kt_integrate_tsdf(self.h, depth.ptr, cols, rows, C.byref(intr), _fp(volume_size), C.byref(Mat33.from_np(Rcurr_inv)),
                                     _fp(tcurr), tranc_dist, volume.ptr, depth_scaled.ptr, _ip(voxel_wrap), color_volume.ptr,
                                     colors.ptr, nmap_curr.ptr, int(angle_color), N))

    def raycast(self, intr: Intr, Rcurr, tcurr, tranc_dist, volume_size, volume, vmap, nmap, cols, rows, voxel_wrap, vmap_color,
                color_volume, N) -> None:
        _chk(lib().kt_raycast(self.h, C.byref(intr), C.byref(Mat33.from_np(Rcurr)), _fp(tcurr), tranc_dist, _fp(volume_size),
                              volume.ptr, vmap.ptr, nmap.ptr, cols, rows, _ip(voxel_wrap), vmap_color.ptr, color_volume.ptr, N))

    def clear_volume(self, vol, elem_size, N, axis, back, current_wrap, delta_wrap) -> None:
        _chk(lib().kt_clear_volume(self.h, vol.ptr, elem_size, N, axis, int(back), current_wrap, delta_wrap))

    def extract_cloud_slice(self, volume, volume_size, out: DevBuf, cap: int, voxel_wrap, color_volume, minX, maxX, minY, maxY,
                            minZ, maxZ, subsample, real_voxel_wrap, N) -> int:
        n = _sz(0)
        _chk(lib().kt_extract_cloud_slice(self.h, volume.ptr, _fp(volume_size), out.ptr, cap, _ip(voxel_wrap), color_volume.ptr,
                                          minX, maxX, minY, maxY, minZ, maxZ, subsample, _ip(real_voxel_wrap), N, C.byref(n)))
        return int(n.value)


class Tracker:
    """kt_tracker: device-resident KintinuousTracker::processFrame."""

    STAGES = ("pyramid", "odometry", "shift", "integrate", "raycast", "resize", "tsdf23")

    def __init__(self, ctx: Ctx, cfg: TrackerConfig):
        self.ctx, self.cfg = ctx, cfg
        h = _vp()
        _chk(lib().kt_tracker_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self) -> None:
        if self.h:
            lib().kt_tracker_destroy(self.h)
            self.h = None

    def __del__(self):   # (a tracker left open would keep later ones off the level form of the ICP chain: kt_tracker_create)
        try:
            self.close()
        except Exception:
            pass

    def reset(self) -> None:
        _chk(lib().kt_tracker_reset(self.h))

    def process_frame(self, depth_dev, rgb_dev, timestamp: int) -> None:
        d = depth_dev.ptr if isinstance(depth_dev, DevBuf) else int(depth_dev)
        r = rgb_dev.ptr if isinstance(rgb_dev, DevBuf) else int(rgb_dev)
        _chk(lib().kt_tracker_process_frame(self.h, d, r, timestamp))

    def host_times(self, reset: bool = False):
        """(mean seconds per process_frame call, of which waiting for the previous frame's pose)."""
        o = (C.c_double * 2)()
        _chk(lib().kt_tracker_host_times(self.h, o, 1 if reset else 0))
        return float(o[0]), float(o[1])

    def enable_slice_stage(self, on: bool, weight_cull: int = 8, k: int = 20) -> None:
        """CloudSliceProcessor's per-slice stage on the device behind every extraction from now on (kt_tracker_enable_slice_stage)."""
        _chk(lib().kt_tracker_enable_slice_stage(self.h, int(on), int(weight_cull), int(k)))

    def slice_processed(self, i: int):
        """processedCloud of slice i (NPOINT_DTYPE records), or None for a slice extracted while the stage was off"""
        n = C.c_longlong(0)
        _chk(lib().kt_tracker_slice_processed_info(self.h, i, C.byref(n)))
        if n.value < 0:
            return None
        out = np.zeros(max(int(n.value), 1), NPOINT_DTYPE)
        _chk(lib().kt_tracker_slice_processed(self.h, i, out.ctypes.data_as(C.c_void_p)))
        return out[: int(n.value)]

    def pose_log(self, enable: Optional[bool] = None, fetch: bool = False):
        """test hook: switch the log of the poses the frames' set-up kernels saw on / off; fetch=True returns it as float32 [n, 12]"""
        if not fetch:
            _chk(lib().kt_tracker_debug_pose_log(self.h, -1 if enable is None else int(enable), None, 0, None))
            return None
        n = C.c_int(0)
        _chk(lib().kt_tracker_debug_pose_log(self.h, -1 if enable is None else int(enable), None, 0, C.byref(n)))
        out = np.zeros((n.value, 12), np.float32)
        if n.value:
            _chk(lib().kt_tracker_debug_pose_log(self.h, -1, out.ctypes.data_as(_pf), n.value, None))
        return out

    def plan_truth(self, poses12, fr: float, ft: float, theta_fixed: float = 0.0, tau_fixed: float = 0.0, seed: int = 1) -> None:
        """test hook: poses of an identical earlier run as the plan's predictions, offset by fr * theta / ft * tau (kt_tracker_debug_plan_truth)"""
        p = np.ascontiguousarray(poses12, np.float32).reshape(-1, 12)
        _chk(lib().kt_tracker_debug_plan_truth(self.h, p.ctypes.data_as(_pf), len(p), fr, ft, theta_fixed, tau_fixed, seed))

    def plan_stats(self) -> Tuple[int, int]:
        """(frames fused from a task plan made ahead of them, frames whose pose fell outside their plan's margins)"""
        out = (C.c_longlong * 2)()
        _chk(lib().kt_tracker_plan_stats(self.h, out))
        return int(out[0]), int(out[1])

    def odometry_fallbacks(self) -> int:
        """frames whose odometry was re-run in the stepwise form after a hand-off time-out (kt_tracker_odometry_fallbacks)"""
        out = C.c_longlong(0)
        _chk(lib().kt_tracker_odometry_fallbacks(self.h, C.byref(out)))
        return int(out.value)

    def prefetch_frame(self, depth_dev, rgb_dev) -> None:
        """Announce a frame a later process_frame call will receive: its pose-independent stages run on a second stream."""
        d = depth_dev.ptr if isinstance(depth_dev, DevBuf) else int(depth_dev)
        r = rgb_dev.ptr if isinstance(rgb_dev, DevBuf) else int(rgb_dev)
        _chk(lib().kt_tracker_prefetch_frame(self.h, d, r))

    def prefetch_frame_host(self, depth: np.ndarray, rgb: np.ndarray) -> None:
        """Host-frame read-ahead: depth / rgb must be C-contiguous uint16 / uint8 arrays; pass the SAME array objects to
        process_frame_host later (they are matched by address)."""
        assert depth.dtype == np.uint16 and rgb.dtype == np.uint8 and depth.flags.c_contiguous and rgb.flags.c_contiguous
        _chk(lib().kt_tracker_prefetch_frame_host(self.h, depth.ctypes.data, rgb.ctypes.data))

    def process_frame_host(self, depth: np.ndarray, rgb: np.ndarray, timestamp: int) -> None:
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        _chk(lib().kt_tracker_process_frame_host(self.h, depth.ctypes.data, rgb.ctypes.data, timestamp))

    def load_trajectory(self, utimes, pose7) -> None:
        """-p ground truth: utimes[n] uint64, pose7[n, 7] = x y z qx qy qz qw (KintinuousTracker::loadTrajectory)."""
        utimes = np.ascontiguousarray(utimes, dtype=np.uint64)
        pose7 = np.ascontiguousarray(pose7, dtype=np.float32).reshape(-1, 7)
        assert len(utimes) == len(pose7)
        _chk(lib().kt_tracker_load_trajectory(self.h, len(utimes), utimes.ctypes.data, pose7.ctypes.data))

    def finalise(self) -> None:
        _chk(lib().kt_tracker_finalise(self.h))

    def volume_basis(self) -> np.ndarray:
        b = (C.c_float * 3)()
        _chk(lib().kt_tracker_get_volume_basis(self.h, b))
        return np.array(b, dtype=np.float32)

    def pose(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        R, t, g = (C.c_float * 9)(), (C.c_float * 3)(), (C.c_float * 3)()
        _chk(lib().kt_tracker_get_pose(self.h, R, t, g))
        return np.array(R, dtype=np.float32).reshape(3, 3), np.array(t, dtype=np.float32), np.array(g, dtype=np.float32)

    def num_poses(self) -> int:
        return _count(lib().kt_tracker_num_poses(self.h))

    def dense_pose(self, i: int) -> Tuple[int, np.ndarray, bool]:
        ts, p, il = _u64(0), (C.c_float * 16)(), C.c_int(0)
        _chk(lib().kt_tracker_get_dense_pose(self.h, i, C.byref(ts), p, C.byref(il)))
        return ts.value, np.array(p, dtype=np.float32).reshape(4, 4), bool(il.value)

    def voxel_wrap(self) -> np.ndarray:
        w = (C.c_int * 3)()
        _chk(lib().kt_tracker_get_voxel_wrap(self.h, w))
        return np.array(w, dtype=np.int32)

    def num_slices(self) -> int:
        return _count(lib().kt_tracker_num_slices(self.h))

    def slice_info(self, i: int) -> Tuple[int, int]:
        """(number of points, CloudSlice::Dimension) without downloading the points."""
        n, dim = _sz(0), C.c_int(0)
        _chk(lib().kt_tracker_slice_info(self.h, i, C.byref(n), C.byref(dim)))
        return n.value, dim.value

    def slice(self, i: int) -> Tuple[np.ndarray, int]:
        n, dim = _sz(0), C.c_int(0)
        _chk(lib().kt_tracker_slice_info(self.h, i, C.byref(n), C.byref(dim)))
        out = np.zeros(n.value, dtype=POINT_DTYPE)
        if n.value:
            _chk(lib().kt_tracker_slice_points(self.h, i, out.ctypes.data))
        return out, dim.value

    def pr_samples(self):
        """[(utime, trans[3], rot[3,3], pose_index)] of the frames sampled for place recognition, in order."""
        out = []
        for i in range(_count(lib().kt_tracker_num_pr_samples(self.h))):
            ut, tr, ro, pi = _u64(0), (C.c_float * 3)(), (C.c_float * 9)(), C.c_int(0)
            _chk(lib().kt_tracker_pr_sample(self.h, i, C.byref(ut), tr, ro, C.byref(pi)))
            out.append((ut.value, np.array(tr, np.float32), np.array(ro, np.float32).reshape(3, 3), pi.value))
        return out

    def slice_pr_id(self, i: int) -> int:
        v = C.c_int(0)
        _chk(lib().kt_tracker_slice_pr_id(self.h, i, C.byref(v)))
        return v.value

    def volume(self) -> np.ndarray:
        N = self.cfg.N
        return self.ctx.download(lib().kt_tracker_volume(self.h), np.int16, (N, N, N))

    def color_volume(self) -> np.ndarray:
        N = self.cfg.N
        return self.ctx.download(lib().kt_tracker_color_volume(self.h), np.uint8, (N, N, N, 4))

    def vmap_g_prev(self, level: int = 0) -> np.ndarray:
        c, r = self.cfg.cols >> level, self.cfg.rows >> level
        return self.ctx.download(lib().kt_tracker_vmap_g_prev(self.h, level), np.float32, (3 * r, c))

    def nmap_g_prev(self, level: int = 0) -> np.ndarray:
        c, r = self.cfg.cols >> level, self.cfg.rows >> level
        return self.ctx.download(lib().kt_tracker_nmap_g_prev(self.h, level), np.float32, (3 * r, c))

    def trunc_dist(self) -> float:
        return float(lib().kt_tracker_trunc_dist(self.h))

    def enable_profiling(self, mode: int) -> None:
        _chk(lib().kt_tracker_enable_profiling(self.h, mode))

    def stage_ms(self) -> dict:
        ms = (C.c_float * 7)()
        n = (C.c_longlong * 7)()
        _chk(lib().kt_tracker_stage_ms(self.h, ms))
        _chk(lib().kt_tracker_stage_counts(self.h, n))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.STAGES)}

    def enable_counts(self, on: bool) -> None:
        _chk(lib().kt_tracker_enable_counts(self.h, int(on)))

    def last_counts(self) -> Tuple[int, int]:
        U, S = C.c_ulonglong(0), C.c_ulonglong(0)
        _chk(lib().kt_tracker_last_counts(self.h, C.byref(U), C.byref(S)))
        return int(U.value), int(S.value)

    def debug_state(self):
        o = (C.c_float * 29)()
        _chk(lib().kt_tracker_debug_state(self.h, o))
        return [float(v) for v in o]

    def debug_counts(self):
        o = (C.c_uint * 8)()
        _chk(lib().kt_tracker_debug_counts(self.h, o))
        return [int(v) for v in o]

    def export_poses_device(self, k: int, dst_ptr: int) -> None:
        _chk(lib().kt_tracker_export_poses_device(self.h, k, dst_ptr))


# ---- host math of the Gauss-Newton step (kt_host_*; no GPU needed) ------------------------------------------------
def _dp(a):
    return a.ctypes.data_as(_pd)


def host_ldlt_solve6(A, b) -> np.ndarray:
    A = np.ascontiguousarray(A, np.float64).reshape(36)
    b = np.ascontiguousarray(b, np.float64).reshape(6)
    x = np.zeros(6, np.float64)
    _chk(lib().kt_host_ldlt_solve6(_dp(A), _dp(b), _dp(x)))
    return x


def host_rodrigues(r) -> np.ndarray:
    r = np.ascontiguousarray(r, np.float64).reshape(3)
    R = np.zeros(9, np.float64)
    _chk(lib().kt_host_rodrigues(_dp(r), _dp(R)))
    return R.reshape(3, 3)


def host_mat33_inverse(m) -> np.ndarray:
    m = np.ascontiguousarray(m, np.float32).reshape(9)
    o = np.zeros(9, np.float32)
    _chk(lib().kt_host_mat33_inverse(m.ctypes.data_as(_pf), o.ctypes.data_as(_pf)))
    return o.reshape(3, 3)


def host_reposition_cube(R, tlast, volume_size, voxel_size, thresh, basis) -> np.ndarray:
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    t = np.ascontiguousarray(tlast, np.float32)
    v = np.ascontiguousarray(voxel_size, np.float32)
    b = np.ascontiguousarray(basis, np.float32).copy()
    lib().kt_host_reposition_cube(R.ctypes.data_as(_pf), t.ctypes.data_as(_pf), float(volume_size), v.ctypes.data_as(_pf), int(thresh), b.ctypes.data_as(_pf))
    return b


def host_pose_update(x, result_rt, Rprev, tprev):
    """Returns (result_rt', Rcurr, tcurr) -- ICPOdometry.cpp:133-178."""
    x = np.ascontiguousarray(x, np.float64).reshape(6)
    rt = np.array(result_rt, np.float64).reshape(16).copy()
    Rp = np.ascontiguousarray(Rprev, np.float32).reshape(9)
    tp = np.ascontiguousarray(tprev, np.float32).reshape(3)
    Rc, tc = np.zeros(9, np.float32), np.zeros(3, np.float32)
    _chk(lib().kt_host_pose_update(_dp(x), _dp(rt), Rp.ctypes.data_as(_pf), tp.ctypes.data_as(_pf),
                                   Rc.ctypes.data_as(_pf), tc.ctypes.data_as(_pf)))
    return rt.reshape(4, 4), Rc.reshape(3, 3), tc


NPOINT_DTYPE = np.dtype([("xyz", np.float32, 3), ("one", np.float32), ("normal", np.float32, 3), ("zero", np.float32), ("bgra", np.uint8, 4),
                         ("curvature", np.float32), ("pad", np.float32, 2)])
assert NPOINT_DTYPE.itemsize == 48


def host_compute_krk(result_rt, fx, fy, cx, cy):
    """K R K^-1 (3x3 float32) and K t (3) of resultRt^-1, RGBDOdometry.cpp:213-231."""
    rt = np.ascontiguousarray(result_rt, np.float64).reshape(16)
    krk, kt = np.zeros(9, np.float32), np.zeros(3, np.float32)
    _chk(lib().kt_host_compute_krk(_dp(rt), float(fx), float(fy), float(cx), float(cy), krk.ctypes.data_as(_pf), kt.ctypes.data_as(_pf)))
    return krk.reshape(3, 3), kt


def host_trajectory_pose(pose7) -> np.ndarray:
    """{x y z qx qy qz qw} -> (R | t) as 12 floats, KintinuousTracker.cpp:244-256."""
    p = np.ascontiguousarray(pose7, np.float32).reshape(7)
    T = np.zeros(12, np.float32)
    lib().kt_host_trajectory_pose(p.ctypes.data_as(_pf), T.ctypes.data_as(_pf))
    return T


def host_ground_truth_pose(A, B, Rlast, tlast):
    """GroundTruthOdometry::getIncrementalTransformation on explicit state: returns (Rcurr, tcurr)."""
    A, B = np.ascontiguousarray(A, np.float32).reshape(12), np.ascontiguousarray(B, np.float32).reshape(12)
    Rl, tl = np.ascontiguousarray(Rlast, np.float32).reshape(9), np.ascontiguousarray(tlast, np.float32).reshape(3)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib().kt_host_ground_truth_pose(A.ctypes.data_as(_pf), B.ctypes.data_as(_pf), Rl.ctypes.data_as(_pf), tl.ctypes.data_as(_pf),
                                    R.ctypes.data_as(_pf), t.ctypes.data_as(_pf))
    return R.reshape(3, 3), t


def slice_process(ctx: "Ctx", points: np.ndarray, weight_cull: int, leaf: float, k: int = 20) -> np.ndarray:
    """kt_slice_process: CloudSliceProcessor's per-slice stage on a host array of extracted points -> pcl::PointXYZRGBNormal records."""
    points = np.ascontiguousarray(points)
    assert points.dtype == POINT_DTYPE
    out = np.zeros(max(len(points), 1), NPOINT_DTYPE)
    n = C.c_size_t(0)
    _chk(lib().kt_slice_process(ctx.h, points.ctypes.data_as(C.c_void_p), len(points), int(weight_cull), float(leaf), int(k),
                                out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out[: n.value]


def voxel_grid_normal(points: np.ndarray, leaf: float) -> np.ndarray:
    """kt_host_voxel_grid_normal: CloudSliceProcessor::save's final pcl::VoxelGrid<pcl::PointXYZRGBNormal> (host code, no GPU)."""
    points = np.ascontiguousarray(points)
    assert points.dtype == NPOINT_DTYPE
    out = np.zeros(max(len(points), 1), NPOINT_DTYPE)
    n = C.c_size_t(0)
    _chk(lib().kt_host_voxel_grid_normal(points.ctypes.data_as(C.c_void_p), len(points), float(leaf), out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out[: n.value]


def save_pcd(path: str, points: np.ndarray) -> None:
    """kt_host_save_pcd: pcl::io::savePCDFile(path, cloud, true) for pcl::PointXYZRGBNormal records."""
    points = np.ascontiguousarray(points)
    assert points.dtype == NPOINT_DTYPE
    _chk(lib().kt_host_save_pcd(path.encode(), points.ctypes.data_as(C.c_void_p), len(points)))


class Comm:
    """The path's single collective through the C-ABI (kt_comm_*: RCCL all-gather of dense poses).  Rank 0 makes the id, `share` hands
    it to the other ranks (bytes -> bytes; identity for one rank)."""

    def __init__(self, ctx: Ctx, rank: int, nranks: int, share=None):
        ident = (C.c_ubyte * 128)()
        if rank == 0:
            _chk(lib().kt_comm_unique_id(ident))
        if share is not None:
            data = share(bytes(ident))
            ident = (C.c_ubyte * 128)(*data)
        self.h = C.c_void_p()
        self.nranks = nranks
        _chk(lib().kt_comm_init(ctx.h, rank, nranks, ident, C.byref(self.h)))

    def gather_poses(self, trk: "Tracker", k: int) -> np.ndarray:
        out = np.zeros((self.nranks, k, 16), np.float32)
        _chk(lib().kt_pose_gather(self.h, trk.h, k, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def barrier(self) -> None:
        """returns once every rank has called it (kt_comm_barrier: a one-float all-gather on the communicator's stream)"""
        _chk(lib().kt_comm_barrier(self.h))

    def close(self) -> None:
        if self.h:
            lib().kt_comm_destroy(self.h)
            self.h = None
