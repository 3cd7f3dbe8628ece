"""ctypes binding of oracle/_ref/libkt_ref.so -- the reference's OWN kernel sources (frontend/cuda/*.cu) compiled for the
host CPU by oracle/Makefile (target _ref/libkt_ref.so) against the CUDA emulation in oracle/ref_shim/.

TEST INFRASTRUCTURE ONLY: it pins the restatement (oracle/kt_oracle_*.c) to the reference, kernel by kernel
(tests/test_oracle_vs_ref.py), and nothing else may import it.  The library is built in the build container, where
/root/reference exists; on the GPU box the prebuilt file is used (oracle/_ref/ is git-ignored but travels with gpurun).
Function names and argument meaning follow oracle/oracle.py, so a test can run the same call through both.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .oracle import DATATERM_DTYPE, POINT_DTYPE, OIntr, OMat33, _c, _f3, _i3, _p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libkt_ref.so")
REFERENCE_DIR = "/root/reference/src/frontend/cuda"

_lib = None


def available() -> bool:
    return os.path.exists(LIB_PATH) or os.path.isdir(REFERENCE_DIR)


def build(force: bool = False) -> str:
    """make -C oracle _ref/libkt_ref.so (needs /root/reference; otherwise the prebuilt library must be there)."""
    if os.path.isdir(REFERENCE_DIR):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []) + ["_ref/libkt_ref.so"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(LIB_PATH):
        raise RuntimeError("oracle/_ref/libkt_ref.so is missing and /root/reference is not present to build it from")
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.ktref_extract_cloud_slice.restype = C.c_size_t
        assert _lib.ktref_sizeof_dataterm() == DATATERM_DTYPE.itemsize and _lib.ktref_sizeof_point() == POINT_DTYPE.itemsize
        assert _lib.ktref_sizeof_jtj() == 29 * 4
    return _lib


# ---- image-side (bilateral_pyrdown.cu, maps.cu, image_generator.cu) -------------------------------------------------
def bilateral_filter(src):
    src = _c(src, np.uint16)
    dst = np.zeros_like(src)
    lib().ktref_bilateral_filter(_p(src), _p(dst), src.shape[1], src.shape[0])
    return dst


def pyr_down(src):
    src = _c(src, np.uint16)
    dst = np.zeros((src.shape[0] // 2, src.shape[1] // 2), np.uint16)
    lib().ktref_pyr_down(_p(src), src.shape[1], src.shape[0], _p(dst))
    return dst


def create_vmap(intr: OIntr, depth, out=None):
    depth = _c(depth, np.uint16)
    rows, cols = depth.shape
    vmap = np.zeros((3 * rows, cols), np.float32) if out is None else out
    lib().ktref_create_vmap(intr, _p(depth), cols, rows, _p(vmap))
    return vmap


def create_nmap(vmap, out=None):
    vmap = _c(vmap, np.float32)
    nmap = np.zeros_like(vmap) if out is None else out
    lib().ktref_create_nmap(_p(vmap), vmap.shape[1], vmap.shape[0] // 3, _p(nmap))
    return nmap


def transform_maps(vmap, nmap, R, t, vout=None, nout=None):
    vmap, nmap = _c(vmap, np.float32), _c(nmap, np.float32)
    vd = np.zeros_like(vmap) if vout is None else vout
    nd = np.zeros_like(nmap) if nout is None else nout
    lib().ktref_transform_maps(_p(vmap), _p(nmap), vmap.shape[1], vmap.shape[0] // 3, C.byref(OMat33.from_np(R)), _f3(t), _p(vd), _p(nd))
    return vd, nd


def resize_map(inp, normalize: bool, out=None):
    inp = _c(inp, np.float32)
    rows, cols = inp.shape[0] // 3, inp.shape[1]
    o = np.zeros((3 * (rows // 2), cols // 2), np.float32) if out is None else out
    (lib().ktref_resize_nmap if normalize else lib().ktref_resize_vmap)(_p(inp), cols, rows, _p(o))
    return o


def generate_image(vmap, nmap, vmap_color, light_pos, light_number: int = 1):
    vmap, nmap, vmap_color = _c(vmap, np.float32), _c(nmap, np.float32), _c(vmap_color, np.uint8)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    dst, dst_color = np.zeros((rows, cols, 3), np.uint8), np.zeros((rows, cols, 3), np.uint8)
    lib().ktref_generate_image(_p(vmap), _p(nmap), _p(vmap_color), cols, rows, _f3(light_pos), light_number, _p(dst), _p(dst_color))
    return dst, dst_color


def generate_depth(R_inv, t, vmap, nmap, max_depth: float):
    vmap, nmap = _c(vmap, np.float32), _c(nmap, np.float32)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    dst = np.zeros((rows, cols), np.uint16)
    lib().ktref_generate_depth(C.byref(OMat33.from_np(R_inv)), _f3(t), _p(vmap), _p(nmap), cols, rows, _p(dst), C.c_float(max_depth))
    return dst


def depth_to_metres(src, cutoff: int):
    src = _c(src, np.uint16)
    dst = np.zeros(src.shape, np.float32)
    lib().ktref_depth_to_metres(_p(src), _p(dst), src.shape[1], src.shape[0], cutoff)
    return dst


def bgr_to_intensity(rgb):
    rgb = _c(rgb, np.uint8)
    dst = np.zeros(rgb.shape[:2], np.uint8)
    lib().ktref_bgr_to_intensity(_p(rgb), _p(dst), rgb.shape[1], rgb.shape[0])
    return dst


def pyr_down_gauss_f32(src):
    src = _c(src, np.float32)
    dst = np.zeros((src.shape[0] // 2, src.shape[1] // 2), np.float32)
    lib().ktref_pyr_down_gauss_f32(_p(src), src.shape[1], src.shape[0], _p(dst))
    return dst


def pyr_down_gauss_u8(src):
    src = _c(src, np.uint8)
    dst = np.zeros((src.shape[0] // 2, src.shape[1] // 2), np.uint8)
    lib().ktref_pyr_down_gauss_u8(_p(src), src.shape[1], src.shape[0], _p(dst))
    return dst


def derivative_images(src):
    src = _c(src, np.uint8)
    dx, dy = np.zeros(src.shape, np.int16), np.zeros(src.shape, np.int16)
    lib().ktref_derivative_images(_p(src), src.shape[1], src.shape[0], _p(dx), _p(dy))
    return dx, dy


def project_to_cloud(depth, fx, fy, cx, cy, level):
    depth = _c(depth, np.float32)
    rows, cols = depth.shape
    cloud = np.zeros((rows, cols, 3), np.float32)
    lib().ktref_project_to_cloud(_p(depth), cols, rows, _p(cloud), C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), level)
    return cloud


# ---- tracking reductions (reduce.cu); threads / blocks as ICPOdometry.cpp:108-125, RGBDOdometry.cpp:236-307 pass them ---
def icp_step(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr: OIntr, vmap_g_prev, nmap_g_prev, dist_thres, angle_thres,
             threads: int = 128, blocks: int = 64):
    vmap_curr, nmap_curr = _c(vmap_curr, np.float32), _c(nmap_curr, np.float32)
    vmap_g_prev, nmap_g_prev = _c(vmap_g_prev, np.float32), _c(nmap_g_prev, np.float32)
    rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
    A, b, r = (C.c_float * 36)(), (C.c_float * 6)(), (C.c_float * 2)()
    lib().ktref_icp_step(C.byref(OMat33.from_np(Rcurr)), _f3(tcurr), _p(vmap_curr), _p(nmap_curr), C.byref(OMat33.from_np(Rprev_inv)), _f3(tprev),
                         intr, _p(vmap_g_prev), _p(nmap_g_prev), cols, rows, C.c_float(dist_thres), C.c_float(angle_thres), threads, blocks, A, b, r)
    return np.array(A, np.float32).reshape(6, 6), np.array(b, np.float32), np.array(r, np.float32)


def rgb_residual(min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, max_depth_delta, kt, krkinv,
                 threads: int = 128, blocks: int = 256):
    dIdx, dIdy = _c(dIdx, np.int16), _c(dIdy, np.int16)
    last_depth, next_depth = _c(last_depth, np.float32), _c(next_depth, np.float32)
    last_image, next_image = _c(last_image, np.uint8), _c(next_image, np.uint8)
    rows, cols = next_image.shape
    corres = np.zeros((rows, cols), DATATERM_DTYPE)
    sigma, count = C.c_int(0), C.c_int(0)
    lib().ktref_rgb_residual(C.c_float(min_scale), _p(dIdx), _p(dIdy), _p(last_depth), _p(next_depth), _p(last_image), _p(next_image), cols, rows,
                             _p(corres), C.c_float(max_depth_delta), _f3(kt), C.byref(OMat33.from_np(krkinv)), threads, blocks,
                             C.byref(sigma), C.byref(count))
    return corres, sigma.value, count.value


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale, threads: int = 128, blocks: int = 64):
    corres = np.ascontiguousarray(corres)
    cloud, dIdx, dIdy = _c(cloud, np.float32), _c(dIdx, np.int16), _c(dIdy, np.int16)
    rows, cols = dIdx.shape
    A, b = (C.c_float * 36)(), (C.c_float * 6)()
    lib().ktref_rgb_step(_p(corres), C.c_float(sigma), _p(cloud), C.c_float(fx), C.c_float(fy), _p(dIdx), _p(dIdy), C.c_float(sobel_scale), cols,
                         rows, threads, blocks, A, b)
    return np.array(A, np.float32).reshape(6, 6), np.array(b, np.float32)


# ---- volume (tsdf_volume.cu, ray_caster.cu, extract.cu) -----------------------------------------------------------------
def init_volume(N: int):
    vol = np.full((N, N, N), 0x5A5A, np.int16)
    lib().ktref_init_volume(_p(vol), N)
    return vol


def init_color_volume(N: int):
    vol = np.full((N, N, N, 4), 0x5A, np.uint8)
    lib().ktref_init_color_volume(_p(vol), N)
    return vol


def integrate_tsdf(depth, intr: OIntr, volume_size, Rcurr_inv, tcurr, tranc_dist, volume, voxel_wrap, color_volume, colors, nmap_curr,
                   angle_color: bool):
    """In place on volume (int16 [N,N,N]) and color_volume (uint8 [N,N,N,4]); returns depth_scaled."""
    depth, colors, nmap_curr = _c(depth, np.uint16), _c(colors, np.uint8), _c(nmap_curr, np.float32)
    assert volume.flags.c_contiguous and color_volume.flags.c_contiguous
    rows, cols = depth.shape
    scaled = np.zeros((rows, cols), np.float32)
    lib().ktref_integrate_tsdf(_p(depth), cols, rows, intr, _f3(volume_size), C.byref(OMat33.from_np(Rcurr_inv)), _f3(tcurr), C.c_float(tranc_dist),
                               _p(volume), _p(scaled), _i3(voxel_wrap), _p(color_volume), _p(colors), _p(nmap_curr), int(angle_color),
                               volume.shape[0])
    return scaled


def raycast(intr: OIntr, Rcurr, tcurr, tranc_dist, volume_size, volume, vmap, nmap, voxel_wrap, vmap_color, color_volume) -> None:
    """In place on vmap, nmap ([3*rows, cols] float32) and vmap_color ([rows, cols, 4] uint8)."""
    assert vmap.flags.c_contiguous and nmap.flags.c_contiguous and vmap_color.flags.c_contiguous
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    lib().ktref_raycast(intr, C.byref(OMat33.from_np(Rcurr)), _f3(tcurr), C.c_float(tranc_dist), _f3(volume_size), _p(volume), _p(vmap), _p(nmap),
                        cols, rows, _i3(voxel_wrap), _p(vmap_color), _p(color_volume), volume.shape[0])


def clear_volume(vol, axis: int, back: bool, current_wrap: int, delta_wrap: int) -> None:
    lib().ktref_clear_volume(_p(vol), 2 if vol.dtype == np.int16 else 4, vol.shape[0], axis, int(back), current_wrap, delta_wrap)


def extract_cloud_slice(volume, volume_size, cap, voxel_wrap, color_volume, minX, maxX, minY, maxY, minZ, maxZ, subsample, real_wrap):
    out = np.zeros(cap, POINT_DTYPE)
    n = lib().ktref_extract_cloud_slice(_p(volume), _f3(volume_size), _p(out), C.c_size_t(cap), _i3(voxel_wrap), _p(color_volume), minX, maxX, minY,
                                        maxY, minZ, maxZ, subsample, _i3(real_wrap), volume.shape[0])
    return out[: int(n)]
