// Shared between kt_image.hip (kt_build_pyramid) and kt_volume.hip (the tracker's fused frame preparation): the pyramid's per-pixel bodies.
#pragma once
#include "kt_common.hpp"

struct kt_pyr_args {
    const uint16_t* d0;
    uint16_t* d[3];          // depth levels 1..3
    float* vmap[4]; float* nmap[4];
    int cols, rows;          // level 0
    float fx_inv[4], fy_inv[4], cx[4], cy[4];
};

// pyrDownGaussKernel body (bilateral_pyrdown.cu:101-136) reading the source level from an LDS tile whose (0,0) is the
// source pixel (tox, toy); returns -1 for destinations outside the destination image.  The reference clips its 5 x 5 window to the source image
// (x_mi .. x_ma); here a tile cell outside the source image holds -1 and is skipped -- the same taps in the same order -- so the loop has
// constant bounds, unrolls, and its 25 LDS reads are in flight together (round 6: the variable-bound loop was a chain of 25 read latencies).
__device__ __forceinline__ int kt_pyr_px(const int* __restrict__ tile, int tw, int tox, int toy, int scols, int srows, int x, int y)
{
    const int dcols = scols / 2, drows = srows / 2;
    if (x < 0 || y < 0 || x >= dcols || y >= drows) return -1;
    const float sigma_color = 30.0f;
    const int* __restrict__ p = tile + (2 * y - toy) * tw + (2 * x - tox);
    const int center = p[0];
    float sum = 0, wall = 0;
#pragma unroll
    for (int yi = -2; yi <= 2; ++yi)
#pragma unroll
        for (int xi = -2; xi <= 2; ++xi) {
            const int val = p[yi * tw + xi];
            if (val >= 0 && (float)abs(val - center) < 3 * sigma_color) {
                const int axi = abs(xi), ayi = abs(yi);
                const float wx = axi == 0 ? 0.375f : (axi == 1 ? 0.25f : 0.0625f);
                const float wy = ayi == 0 ? 0.375f : (ayi == 1 ? 0.25f : 0.0625f);
                sum = __builtin_fmaf((float)val * wx, wy, sum);
                wall = __builtin_fmaf(wx, wy, wall);
            }
        }
    return (int)(uint16_t)kt_f2i_rz(sum / wall);
}

// computeVmapKernel + computeNmapKernel (maps.cu:56-120) for pixel (u, v) of a level whose depth sits in an LDS tile
__device__ __forceinline__ void kt_emit_maps(const int* __restrict__ tile, int tw, int tox, int toy, int cols, int rows, int u, int v,
                                             float fx_inv, float fy_inv, float cx, float cy, float* __restrict__ vmap, float* __restrict__ nmap)
{
    if (u >= cols || v >= rows) return;
    const float z00 = (float)tile[(v - toy) * tw + (u - tox)] / 1000.f;
    f3 v00 = {kt_nan(), 0.f, 0.f};
    if (z00 != 0) {
        v00 = {z00 * ((float)u - cx) * fx_inv, z00 * ((float)v - cy) * fy_inv, z00};
        vmap[v * cols + u] = v00.x;
        vmap[(v + rows) * cols + u] = v00.y;
        vmap[(v + 2 * rows) * cols + u] = v00.z;
    } else
        vmap[v * cols + u] = kt_nan();
    if (u == cols - 1 || v == rows - 1) { nmap[v * cols + u] = kt_nan(); return; }
    const float z01 = (float)tile[(v - toy) * tw + (u + 1 - tox)] / 1000.f;
    const float z10 = (float)tile[(v + 1 - toy) * tw + (u - tox)] / 1000.f;
    if (z00 != 0 && z01 != 0 && z10 != 0) {
        const f3 v01 = {z01 * ((float)(u + 1) - cx) * fx_inv, z01 * ((float)v - cy) * fy_inv, z01};
        const f3 v10 = {z10 * ((float)u - cx) * fx_inv, z10 * ((float)(v + 1) - cy) * fy_inv, z10};
        const f3 r = kt_normalized(kt_cross(kt_sub(v01, v00), kt_sub(v10, v00)));
        nmap[v * cols + u] = r.x;
        nmap[(v + rows) * cols + u] = r.y;
        nmap[(v + 2 * rows) * cols + u] = r.z;
    } else
        nmap[v * cols + u] = kt_nan();
}

// S = side of the level-3 tile a workgroup owns; tile widths: level 3: S + 1, level 2: 2 S + 5, level 1: 4 S + 13
template <int S>
__device__ __forceinline__ void kt_pyramid23_block(const kt_pyr_args& a, int block_x, int block_y, int tid)
{
    constexpr int T3 = S + 1, T2 = 2 * S + 5, T1 = 4 * S + 13;
    __shared__ int t1[T1 * T1], t2[T2 * T2], t3[T3 * T3];
    const int o3x = block_x * S, o3y = block_y * S;
    const int o2x = 2 * o3x, o2y = 2 * o3y, o1x = 4 * o3x, o1y = 4 * o3y;
    const int c1 = a.cols / 2, r1 = a.rows / 2, c2 = c1 / 2, r2 = r1 / 2, c3 = c2 / 2, r3 = r2 / 2;
    const int t1x = o1x - 6, t1y = o1y - 6, t2x = o2x - 2, t2y = o2y - 2, t3x = o3x, t3y = o3y;
    for (int i = tid; i < T1 * T1; i += 256) {
        const int ly = i / T1, lx = i - ly * T1;
        const int gx = t1x + lx, gy = t1y + ly;
        t1[i] = (gx >= 0 && gy >= 0 && gx < c1 && gy < r1) ? (int)a.d[0][gy * c1 + gx] : -1;
    }
    __syncthreads();
    for (int i = tid; i < T2 * T2; i += 256) {
        const int ly = i / T2, lx = i - ly * T2;
        t2[i] = kt_pyr_px(t1, T1, t1x, t1y, c1, r1, t2x + lx, t2y + ly);
    }
    __syncthreads();
    if (tid < T3 * T3) {
        const int ly = tid / T3, lx = tid - ly * T3;
        t3[tid] = kt_pyr_px(t2, T2, t2x, t2y, c2, r2, t3x + lx, t3y + ly);
    }
    __syncthreads();
    if (tid < 4 * S * S) {  // level 2: 2 S x 2 S
        const int ly = tid / (2 * S), lx = tid - ly * (2 * S);
        const int u = o2x + lx, v = o2y + ly;
        if (u < c2 && v < r2) a.d[1][v * c2 + u] = (uint16_t)t2[(v - t2y) * T2 + (u - t2x)];
        kt_emit_maps(t2, T2, t2x, t2y, c2, r2, u, v, a.fx_inv[2], a.fy_inv[2], a.cx[2], a.cy[2], a.vmap[2], a.nmap[2]);
    }
    const int q3 = 255 - tid;   // level 3 on the last wave
    if (q3 < S * S) {
        const int ly = q3 / S, lx = q3 - ly * S;
        const int u = o3x + lx, v = o3y + ly;
        if (u < c3 && v < r3) a.d[2][v * c3 + u] = (uint16_t)t3[(v - t3y) * T3 + (u - t3x)];
        kt_emit_maps(t3, T3, t3x, t3y, c3, r3, u, v, a.fx_inv[3], a.fy_inv[3], a.cx[3], a.cy[3], a.vmap[3], a.nmap[3]);
    }
}

