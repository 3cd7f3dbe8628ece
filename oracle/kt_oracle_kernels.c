/*
 * kt_oracle_kernels.c -- CPU restatement of the reference's device kernels.
 * TEST INFRASTRUCTURE ONLY (see kt_oracle.h).  Pinned bit for bit against the reference's own .cu files built for the host
 * (oracle/_ref, tests/test_oracle_vs_ref.py).
 * Every function cites the reference file:line it follows (paths relative to /root/reference/src/).
 * Build: gcc -O2 -ffp-contract=off -fopenmp (see oracle/Makefile).  Never build with -ffast-math.
 */
#include "kt_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define KTO_DIVISOR 32767               /* internal.h:237 */
#define KTO_RGB_VIEW_ANGLE_WEIGHT 0.75f /* internal.h:241 */
#define KTO_MAX_WEIGHT 128.0f           /* tsdf_volume.cu:481-488 */

static inline float kto_nan(void) { union { uint32_t u; float f; } c; c.u = 0x7fffffffu; return c.f; } /* limits.hpp */
static inline int kto_isnan(float x) { return x != x; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------------
 * CUDA float->int conversions: saturating, NaN -> 0.
 * ---------------------------------------------------------------------------------------------- */
int kto_f2i_rn(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT_MAX;
    if (x <= -2147483648.0f) return INT_MIN;
    return (int)rintf(x); /* default rounding mode: nearest-even */
}
int kto_f2i_rz(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT_MAX;
    if (x <= -2147483648.0f) return INT_MIN;
    return (int)truncf(x);
}
int kto_f2i_rd(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT_MAX;
    if (x <= -2147483648.0f) return INT_MIN;
    return (int)floorf(x);
}
/* float -> uchar as cvt.rzi.u8.f32: truncate, saturate to [0,255], NaN -> 0 */
static inline uint8_t f2u8_rz(float x)
{
    if (x != x) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)(int)x;
}
/* float -> short as cvt.rzi.s16.f32 */
static inline int16_t f2s16_rz(float x)
{
    if (x != x) return 0;
    if (x <= -32768.0f) return -32768;
    if (x >= 32767.0f) return 32767;
    return (int16_t)(int)x;
}

/* Restatement of __expf (bilateral_pyrdown.cu:89): exp(x) = 2^n * e^r, n = rint(x*log2e),
 * r = x - n*ln2 (two-step Cody-Waite), e^r by the cephes degree-6 polynomial, every step an explicit
 * fmaf so that the HIP kernel can repeat it bit for bit.  Results below 2^-125 flush to 0 (the reference
 * is built with --ftz=true) and from 2^128 up the result is +inf, like ex2.approx: the argument is positive when the
 * int product (value - tmp)^2 of bilateralKernel wraps (depth differences > 46340 mm; found by oracle/_ref). */
float kto_expf(float x)
{
    float t = x * 1.44269504088896341f;
    if (!(t >= -125.0f)) return 0.0f; /* also catches NaN */
    if (t >= 128.0f) return INFINITY;
    float n = rintf(t);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float e = fmaf(p, r2, r) + 1.0f;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return e * s.f;
}

/* vector_math.hpp:53-61, with nvcc -fmad contraction written out (LLVM rule, see kt_oracle.h) */
static inline float dot3(const float a[3], const float b[3])
{
    return fmaf(a[2], b[2], fmaf(a[0], b[0], a[1] * b[1]));
}
static inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
/* vector_math.hpp:98-102 */
static inline void mat33_mul(const kto_mat33* m, const float v[3], float o[3])
{
    float r0 = dot3(&m->m[0], v), r1 = dot3(&m->m[3], v), r2 = dot3(&m->m[6], v);
    o[0] = r0; o[1] = r1; o[2] = r2;
}
/* normalized(): v * rsqrtf(dot(v,v)), rsqrtf restated as 1/sqrtf (vector_math.hpp:83-91) */
static inline void normalize3(float v[3])
{
    float inv = 1.0f / sqrtf(dot3(v, v));
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

/* ================================================================================================
 * a1  bilateralFilter -> bilateralKernel            bilateral_pyrdown.cu:59-99, 332-342
 * ============================================================================================== */
void kto_bilateral_filter(const uint16_t* src, uint16_t* dst, int cols, int rows)
{
    const float sigma_color = 30.0f, sigma_space = 4.5f;                       /* :56-57 */
    const float sigma_space2_inv_half = 0.5f / (sigma_space * sigma_space);     /* :338 */
    const float sigma_color2_inv_half = 0.5f / (sigma_color * sigma_color);
    const int R = 6, D = R * 2 + 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int value = src[y * cols + x];
            int tx = imin(x - D / 2 + D, cols - 1); /* exclusive, clipped to cols-1: quirk A.5 */
            int ty = imin(y - D / 2 + D, rows - 1);
            float sum1 = 0, sum2 = 0;
            for (int cy = imax(y - D / 2, 0); cy < ty; ++cy)
                for (int cx = imax(x - D / 2, 0); cx < tx; ++cx) {
                    int tmp = src[cy * cols + cx];
                    float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                    float color2 = (float)(int)((unsigned)(value - tmp) * (unsigned)(value - tmp));
                    float weight = kto_expf(-fmaf(space2, sigma_space2_inv_half, color2 * sigma_color2_inv_half));
                    sum1 = fmaf((float)tmp, weight, sum1);
                    sum2 += weight;
                }
            int res = kto_f2i_rn(sum1 / sum2);
            dst[y * cols + x] = (uint16_t)imax(0, imin(res, 32767));
        }
}

/* ================================================================================================
 * a2  pyrDown -> pyrDownGaussKernel                 bilateral_pyrdown.cu:101-136, 344-354
 * ============================================================================================== */
void kto_pyr_down(const uint16_t* src, int scols, int srows, uint16_t* dst)
{
    const int dcols = scols / 2, drows = srows / 2, D = 5;
    const float sigma_color = 30.0f;
    const float weights[3] = {0.375f, 0.25f, 0.0625f};
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            int center = src[(2 * y) * scols + 2 * x];
            int x_mi = imax(0, 2 * x - D / 2) - 2 * x;
            int y_mi = imax(0, 2 * y - D / 2) - 2 * y;
            int x_ma = imin(scols, 2 * x - D / 2 + D) - 2 * x;
            int y_ma = imin(srows, 2 * y - D / 2 + D) - 2 * y;
            float sum = 0, wall = 0;
            for (int yi = y_mi; yi < y_ma; ++yi)
                for (int xi = x_mi; xi < x_ma; ++xi) {
                    int val = src[(2 * y + yi) * scols + 2 * x + xi];
                    if ((float)abs(val - center) < 3 * sigma_color) {
                        float wx = weights[abs(xi)], wy = weights[abs(yi)];
                        sum = fmaf((float)val * wx, wy, sum);
                        wall = fmaf(wx, wy, wall);
                    }
                }
            dst[y * dcols + x] = (uint16_t)kto_f2i_rz(sum / wall); /* static_cast<int>: truncation, quirk A.6 */
        }
}

/* ================================================================================================
 * a3  createVMap -> computeVmapKernel               maps.cu:56-80, 122-137
 * ============================================================================================== */
void kto_create_vmap(kto_intr intr, const uint16_t* depth, int cols, int rows, float* vmap)
{
    const float fx_inv = 1.f / intr.fx, fy_inv = 1.f / intr.fy, cx = intr.cx, cy = intr.cy;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            float z = (float)depth[v * cols + u] / 1000.f;
            if (z != 0) {
                vmap[v * cols + u] = z * ((float)u - cx) * fx_inv;
                vmap[(v + rows) * cols + u] = z * ((float)v - cy) * fy_inv;
                vmap[(v + 2 * rows) * cols + u] = z;
            } else
                vmap[v * cols + u] = kto_nan(); /* x plane only: quirk A.7 */
        }
}

/* ================================================================================================
 * a4  createNMap -> computeNmapKernel               maps.cu:82-120, 139-154
 * ============================================================================================== */
void kto_create_nmap(const float* vmap, int cols, int rows, float* nmap)
{
#pragma omp parallel for schedule(static)
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            if (u == cols - 1 || v == rows - 1) { nmap[v * cols + u] = kto_nan(); continue; }
            float v00[3], v01[3], v10[3];
            v00[0] = vmap[v * cols + u];
            v01[0] = vmap[v * cols + u + 1];
            v10[0] = vmap[(v + 1) * cols + u];
            if (!kto_isnan(v00[0]) && !kto_isnan(v01[0]) && !kto_isnan(v10[0])) {
                v00[1] = vmap[(v + rows) * cols + u];
                v01[1] = vmap[(v + rows) * cols + u + 1];
                v10[1] = vmap[(v + 1 + rows) * cols + u];
                v00[2] = vmap[(v + 2 * rows) * cols + u];
                v01[2] = vmap[(v + 2 * rows) * cols + u + 1];
                v10[2] = vmap[(v + 1 + 2 * rows) * cols + u];
                float a[3] = {v01[0] - v00[0], v01[1] - v00[1], v01[2] - v00[2]};
                float b[3] = {v10[0] - v00[0], v10[1] - v00[1], v10[2] - v00[2]};
                float r[3];
                cross3(a, b, r);
                normalize3(r);
                nmap[v * cols + u] = r[0];
                nmap[(v + rows) * cols + u] = r[1];
                nmap[(v + 2 * rows) * cols + u] = r[2];
            } else
                nmap[v * cols + u] = kto_nan();
        }
}

/* ================================================================================================
 * view products (internal.h:435-442): generateImage -> generateImageKernel (image_generator.cu:56-179),
 * generateDepth -> generateDepthKernel (image_generator.cu:181-219).  Not on the tracked path: they complete the internal.h surface
 * (KintinuousTracker::getImage / getModelDepth, KintinuousTracker.cpp:960-981).
 * ============================================================================================== */
static void heat_map_color(float value, int* red, int* green, int* blue)
{
    static const float color[4][3] = {{0, 0, 1}, {0, 1, 0}, {1, 1, 0}, {1, 0, 0}};
    int idx1, idx2;
    float fract = 0;
    if (value <= 0) idx1 = idx2 = 0;
    else if (value >= 1) idx1 = idx2 = 3;
    else {
        value = value * 3;
        idx1 = (int)floorf(value);
        idx2 = idx1 + 1;
        fract = value - (float)idx1;
    }
    *red = kto_f2i_rz(fmaf(color[idx2][0] - color[idx1][0], fract, color[idx1][0]) * 235.0f);
    *green = kto_f2i_rz(fmaf(color[idx2][1] - color[idx1][1], fract, color[idx1][1]) * 235.0f);
    *blue = kto_f2i_rz(fmaf(color[idx2][2] - color[idx1][2], fract, color[idx1][2]) * 235.0f);
}

void kto_generate_image(const float* vmap, const float* nmap, const uint8_t* vmap_curr_color, int cols, int rows,
                        const float light_pos[3], int light_number, uint8_t* dst, uint8_t* dst_color)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float vx = vmap[y * cols + x], nx = nmap[y * cols + x];
            const uint8_t* cc = &vmap_curr_color[4 * (y * cols + x)];
            const int ok = !kto_isnan(vx) && !kto_isnan(nx);
            uint8_t c3[3] = {0, 0, 0};
            if (ok) { c3[0] = cc[0]; c3[1] = cc[1]; c3[2] = cc[2]; }
            memcpy(&dst_color[3 * (y * cols + x)], c3, 3);
            uint8_t s3[3] = {0, 0, 0};
            if (ok) {
                const float v[3] = {vx, vmap[(y + rows) * cols + x], vmap[(y + 2 * rows) * cols + x]};
                const float n[3] = {nx, nmap[(y + rows) * cols + x], nmap[(y + 2 * rows) * cols + x]};
                float weight = 1.f;
                for (int i = 0; i < light_number; ++i) { /* LightSource holds one position (internal.h:289-293) */
                    float vec[3] = {light_pos[0] - v[0], light_pos[1] - v[1], light_pos[2] - v[2]};
                    normalize3(vec);
                    weight *= fabsf(dot3(vec, n));
                }
                int r, g, b;
                heat_map_color((float)cc[3] / 128.0f, &r, &g, &b);
                s3[0] = (uint8_t)kto_f2i_rz(fmaf((float)b, weight, 20.f));
                s3[1] = (uint8_t)kto_f2i_rz(fmaf((float)g, weight, 20.f));
                s3[2] = (uint8_t)kto_f2i_rz(fmaf((float)r, weight, 20.f));
            }
            memcpy(&dst[3 * (y * cols + x)], s3, 3);
        }
}

void kto_generate_depth(const kto_mat33* R_inv, const float t[3], const float* vmap, const float* nmap, int cols, int rows, uint16_t* dst)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            uint16_t result = 0;
            const float vx = vmap[y * cols + x], nx = nmap[y * cols + x];
            if (!kto_isnan(vx) && !kto_isnan(nx)) {
                const float d[3] = {vx - t[0], vmap[(y + rows) * cols + x] - t[1], vmap[(y + 2 * rows) * cols + x] - t[2]};
                const float v_z = dot3(&R_inv->m[6], d);
                /* static_cast<unsigned short>(float): cvt.rzi.u16.f32 -- truncate, clamp to [0, 65535], NaN -> 0 */
                const float m = v_z * 1000;
                result = (m != m || m <= 0.0f) ? 0 : (m >= 65535.0f ? 65535 : (uint16_t)(int)m);
            }
            dst[y * cols + x] = result;
        }
}

/* ================================================================================================
 * a5  tranformMaps -> tranformMapsKernel            maps.cu:156-223
 * ============================================================================================== */
void kto_transform_maps(const float* vmap_src, const float* nmap_src, int cols, int rows,
                        const kto_mat33* R, const float t[3], float* vmap_dst, float* nmap_dst)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float vs[3], vd0 = kto_nan();
            vs[0] = vmap_src[y * cols + x];
            if (!kto_isnan(vs[0])) {
                vs[1] = vmap_src[(y + rows) * cols + x];
                vs[2] = vmap_src[(y + 2 * rows) * cols + x];
                float o[3];
                mat33_mul(R, vs, o);
                vd0 = o[0] + t[0];
                vmap_dst[(y + rows) * cols + x] = o[1] + t[1];
                vmap_dst[(y + 2 * rows) * cols + x] = o[2] + t[2];
            }
            vmap_dst[y * cols + x] = vd0;
            float ns[3], nd0 = kto_nan();
            ns[0] = nmap_src[y * cols + x];
            if (!kto_isnan(ns[0])) {
                ns[1] = nmap_src[(y + rows) * cols + x];
                ns[2] = nmap_src[(y + 2 * rows) * cols + x];
                float o[3];
                mat33_mul(R, ns, o);
                nd0 = o[0];
                nmap_dst[(y + rows) * cols + x] = o[1];
                nmap_dst[(y + 2 * rows) * cols + x] = o[2];
            }
            nmap_dst[y * cols + x] = nd0;
        }
}

/* ================================================================================================
 * a13 resizeVMap / resizeNMap -> resizeMapKernel<normalize>   maps.cu:225-308
 * ============================================================================================== */
static void resize_map(const float* in, int in_cols, int in_rows, float* out, int normalize)
{
    const int dcols = in_cols / 2, drows = in_rows / 2, srows = in_rows;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            int xs = x * 2, ys = y * 2;
            float x00 = in[(ys + 0) * in_cols + xs + 0], x01 = in[(ys + 0) * in_cols + xs + 1];
            float x10 = in[(ys + 1) * in_cols + xs + 0], x11 = in[(ys + 1) * in_cols + xs + 1];
            if (kto_isnan(x00) || kto_isnan(x01) || kto_isnan(x10) || kto_isnan(x11)) {
                out[y * dcols + x] = kto_nan();
                continue;
            }
            float n[3];
            n[0] = (x00 + x01 + x10 + x11) / 4;
            float y00 = in[(ys + srows + 0) * in_cols + xs + 0], y01 = in[(ys + srows + 0) * in_cols + xs + 1];
            float y10 = in[(ys + srows + 1) * in_cols + xs + 0], y11 = in[(ys + srows + 1) * in_cols + xs + 1];
            n[1] = (y00 + y01 + y10 + y11) / 4;
            float z00 = in[(ys + 2 * srows + 0) * in_cols + xs + 0], z01 = in[(ys + 2 * srows + 0) * in_cols + xs + 1];
            float z10 = in[(ys + 2 * srows + 1) * in_cols + xs + 0], z11 = in[(ys + 2 * srows + 1) * in_cols + xs + 1];
            n[2] = (z00 + z01 + z10 + z11) / 4;
            if (normalize) normalize3(n);
            out[y * dcols + x] = n[0];
            out[(y + drows) * dcols + x] = n[1];
            out[(y + 2 * drows) * dcols + x] = n[2];
        }
}
void kto_resize_vmap(const float* in, int in_cols, int in_rows, float* out) { resize_map(in, in_cols, in_rows, out, 0); }
void kto_resize_nmap(const float* in, int in_cols, int in_rows, float* out) { resize_map(in, in_cols, in_rows, out, 1); }

/* ================================================================================================
 * a10 helpers
 * ============================================================================================== */
/* shortDepthToMetres -> short2FloatKernel           bilateral_pyrdown.cu:231-242, 404-411 */
void kto_depth_to_metres(const uint16_t* src, float* dst, int cols, int rows, int cutoff)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < cols * rows; ++i) {
        int s = src[i];
        dst[i] = (s > cutoff || s <= 0) ? kto_nan() : ((float)s) / 1000.0f;
    }
}
/* imageBGRToIntensity -> bgr2IntensityKernel        bilateral_pyrdown.cu:244-258, 413-420.
 * PixelRGB fields r,g,b are bytes 0,1,2 of the rgb24 pixel (internal.h:151-154). */
void kto_bgr_to_intensity(const uint8_t* src, uint8_t* dst, int cols, int rows)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < cols * rows; ++i) {
        float r = (float)src[3 * i + 0], g = (float)src[3 * i + 1], b = (float)src[3 * i + 2];
        int value = kto_f2i_rz(fmaf(g, 0.587f, fmaf(r, 0.114f, b * 0.299f)));
        dst[i] = (uint8_t)value;
    }
}
static const float kGauss5x5[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};
/* pyrDownGaussF -> pyrDownKernelGaussF              bilateral_pyrdown.cu:199-229, 356-377 */
void kto_pyr_down_gauss_f32(const float* src, int scols, int srows, float* dst)
{
    const int dcols = scols / 2, drows = srows / 2, D = 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            int tx = imin(2 * x - D / 2 + D, scols - 1);
            int ty = imin(2 * y - D / 2 + D, srows - 1);
            float sum = 0;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    float s = src[cy * scols + cx];
                    if (!kto_isnan(s)) {
                        float g = kGauss5x5[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum = fmaf(s, g, sum);
                        count = kto_f2i_rz((float)count + g); /* int += float */
                    }
                }
            dst[y * dcols + x] = sum / (float)count;
        }
}
/* pyrDownUcharGauss -> pyrDownKernelIntensityGauss  bilateral_pyrdown.cu:171-197, 379-402 */
void kto_pyr_down_gauss_u8(const uint8_t* src, int scols, int srows, uint8_t* dst)
{
    const int dcols = scols / 2, drows = srows / 2, D = 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            int tx = imin(2 * x - D / 2 + D, scols - 1);
            int ty = imin(2 * y - D / 2 + D, srows - 1);
            float sum = 0;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    float g = kGauss5x5[(ty - cy - 1) * 5 + (tx - cx - 1)];
                    sum = fmaf((float)src[cy * scols + cx], g, sum);
                    count = kto_f2i_rz((float)count + g);
                }
            dst[y * dcols + x] = f2u8_rz(sum / (float)count);
        }
}
/* computeDerivativeImages -> applyKernel            bilateral_pyrdown.cu:271-330 */
void kto_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy)
{
    /* double literals narrowed to float, as in the reference's float gsx3x3[9] = {0.52201, ...} */
    static const float gsx[9] = {(float)0.52201, (float)0.00000, (float)-0.52201, (float)0.79451, (float)-0.00000,
                                 (float)-0.79451, (float)0.52201, (float)0.00000, (float)-0.52201};
    static const float gsy[9] = {(float)0.52201, (float)0.79451, (float)0.52201, (float)0.00000, (float)0.00000,
                                 (float)0.00000, (float)-0.52201, (float)-0.79451, (float)-0.52201};
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float dxVal = 0, dyVal = 0;
            int k = 8; /* border taps consume the kernel from index 8 downwards: quirk */
            for (int j = imax(y - 1, 0); j <= imin(y + 1, rows - 1); j++)
                for (int i = imax(x - 1, 0); i <= imin(x + 1, cols - 1); i++) {
                    float s = (float)src[j * cols + i];
                    dxVal = fmaf(s, gsx[k], dxVal);
                    dyVal = fmaf(s, gsy[k], dyVal);
                    --k;
                }
            dx[y * cols + x] = f2s16_rz(dxVal);
            dy[y * cols + x] = f2s16_rz(dyVal);
        }
}
/* projectToPointCloud -> projectPointsKernel        maps.cu:310-344 */
void kto_project_to_cloud(const float* depth, int cols, int rows, float* cloud, double fx, double fy, double cx,
                          double cy, int level)
{
    const int div = 1 << level; /* IntrDoublePrecision::operator() internal.h:268-272 */
    const double lfx = fx / div, lfy = fy / div, lcx = cx / div, lcy = cy / div;
    const double invFx = 1.0f / lfx, invFy = 1.0f / lfy;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float z = depth[y * cols + x];
            float* c = &cloud[3 * (y * cols + x)];
            c[0] = (float)((x - lcx) * z * invFx);
            c[1] = (float)((y - lcy) * z * invFy);
            c[2] = z;
        }
}

/* ================================================================================================
 * 29-float reduction in the reference's exact summation order   reduce.cu:89-184
 *   grid `blocks` x `threads` grid-stride partial sums -> warp shuffle tree (32 lanes) -> shared[32]
 *   -> first-warp tree -> reduceSum<<<1,512>>> over the block partials (host MAX_THREADS == 512).
 * `vals` = per-element 29-vectors, n elements.  order 1: plain double accumulation of the same floats.
 * ============================================================================================== */
static void warp_tree(float (*w)[29], int lanes_base)
{
    /* val += __shfl_down(val, offset) for offset 16,8,4,2,1; only the lanes feeding lane 0 matter */
    for (int off = 16; off > 0; off /= 2)
        for (int l = 0; l < off; ++l)
            for (int k = 0; k < 29; ++k) w[lanes_base + l][k] += w[lanes_base + l + off][k];
}
static void block_reduce(float (*thr)[29], int nthreads, float out[29])
{
    /* blockReduceSum reduce.cu:131-164 */
    int nwarps = nthreads / 32;
    float shared[32][29];
    memset(shared, 0, sizeof(shared));
    for (int w = 0; w < nwarps; ++w) {
        warp_tree(thr, w * 32);
        memcpy(shared[w], thr[w * 32], sizeof(float) * 29);
    }
    /* threadIdx.x < blockDim.x / warpSize ? shared[lane] : zero; rows >= nwarps are already zero */
    warp_tree(shared, 0);
    memcpy(out, shared[0], sizeof(float) * 29);
}
/* `rows` = per-element 8-vectors {row[0..6], found}, n elements.  The per-thread accumulation is
 * `sum.add(getProducts(i))` (reduce.cu:322-327 / :531-536): after inlining every one of the 28 products row[i] * row[j]
 * has the accumulator as its only user, so `acc += row[i] * row[j]` contracts to fma(row[i], row[j], acc) under nvcc's
 * default -fmad=true (and under clang -ffp-contract=fast, which is how oracle/_ref pins it); inliers is a plain add. */
static void reduce29(const float* rows, int n, int threads, int blocks, int order, float out[29])
{
    if (order == 1) {
        double acc[29];
        for (int k = 0; k < 29; ++k) acc[k] = 0;
        for (int i = 0; i < n; ++i) {
            const float* r = &rows[(size_t)i * 8];
            int s = 0;
            for (int a = 0; a < 7; ++a)
                for (int b = a; b < 7; ++b) acc[s++] += (double)r[a] * (double)r[b];
            acc[28] += (double)r[7];
        }
        for (int k = 0; k < 29; ++k) out[k] = (float)acc[k];
        return;
    }
    const int T = threads * blocks;
    float(*thr)[29] = calloc((size_t)T, sizeof(*thr));
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t)
        for (int i = t; i < n; i += T) {
            const float* r = &rows[(size_t)i * 8];
            int s = 0;
            for (int a = 0; a < 7; ++a)
                for (int b = a; b < 7; ++b, ++s) thr[t][s] = fmaf(r[a], r[b], thr[t][s]);
            thr[t][28] += r[7];
        }
    float(*part)[29] = calloc(512, sizeof(*part)); /* reduceSum<<<1, 512>>>: thread i<blocks loads in[i] */
    for (int b = 0; b < blocks; ++b) block_reduce(&thr[b * threads], threads, part[b]);
    block_reduce(part, 512, out);
    free(thr);
    free(part);
}
/* host unpack, reduce.cu:401-418 */
static void unpack29(const float h[29], float A[36], float b[6], float residual[2])
{
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            float value = h[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    if (residual) { residual[0] = h[27]; residual[1] = h[28]; }
}
static inline void store_row8(const float row[7], float found, float* v)
{
    for (int i = 0; i < 7; ++i) v[i] = row[i];
    v[7] = found;
}

/* ================================================================================================
 * a6  icpStep -> icpKernel + reduceSum              reduce.cu:186-419
 * ============================================================================================== */
void kto_icp_step(const kto_mat33* Rcurr, const float tcurr[3], const float* vmap_curr, const float* nmap_curr,
                  const kto_mat33* Rprev_inv, const float tprev[3], kto_intr intr, const float* vmap_g_prev,
                  const float* nmap_g_prev, int cols, int rows, float dist_thres, float angle_thres, int order,
                  float A[36], float b[6], float residual[2])
{
    const int n = cols * rows;
    float* vals = malloc((size_t)n * 8 * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int y = i / cols, x = i - y * cols;
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        int found = 0;
        /* search(): reduce.cu:213-254 */
        float vcurr[3] = {vmap_curr[y * cols + x], vmap_curr[(y + rows) * cols + x], vmap_curr[(y + 2 * rows) * cols + x]};
        float vcurr_g[3], tmp[3], vcurr_cp[3];
        mat33_mul(Rcurr, vcurr, vcurr_g);
        vcurr_g[0] += tcurr[0]; vcurr_g[1] += tcurr[1]; vcurr_g[2] += tcurr[2];
        tmp[0] = vcurr_g[0] - tprev[0]; tmp[1] = vcurr_g[1] - tprev[1]; tmp[2] = vcurr_g[2] - tprev[2];
        mat33_mul(Rprev_inv, tmp, vcurr_cp);
        int ux = kto_f2i_rn(vcurr_cp[0] * intr.fx / vcurr_cp[2] + intr.cx);
        int uy = kto_f2i_rn(vcurr_cp[1] * intr.fy / vcurr_cp[2] + intr.cy);
        if (!(ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp[2] < 0)) {
            float vprev_g[3] = {vmap_g_prev[uy * cols + ux], vmap_g_prev[(uy + rows) * cols + ux], vmap_g_prev[(uy + 2 * rows) * cols + ux]};
            float ncurr[3] = {nmap_curr[y * cols + x], nmap_curr[(y + rows) * cols + x], nmap_curr[(y + 2 * rows) * cols + x]};
            float ncurr_g[3];
            mat33_mul(Rcurr, ncurr, ncurr_g);
            float nprev_g[3] = {nmap_g_prev[uy * cols + ux], nmap_g_prev[(uy + rows) * cols + ux], nmap_g_prev[(uy + 2 * rows) * cols + ux]};
            float dv[3] = {vprev_g[0] - vcurr_g[0], vprev_g[1] - vcurr_g[1], vprev_g[2] - vcurr_g[2]};
            float dist = sqrtf(dot3(dv, dv));
            float cr[3];
            cross3(ncurr_g, nprev_g, cr);
            float sine = sqrtf(dot3(cr, cr));
            found = (sine < angle_thres && dist <= dist_thres && !kto_isnan(ncurr[0]) && !kto_isnan(nprev_g[0]));
            if (found) {
                /* getProducts(): reduce.cu:256-277 */
                float s_cp[3], d_cp[3], n_cp[3], t2[3];
                t2[0] = vcurr_g[0] - tprev[0]; t2[1] = vcurr_g[1] - tprev[1]; t2[2] = vcurr_g[2] - tprev[2];
                mat33_mul(Rprev_inv, t2, s_cp);
                t2[0] = vprev_g[0] - tprev[0]; t2[1] = vprev_g[1] - tprev[1]; t2[2] = vprev_g[2] - tprev[2];
                mat33_mul(Rprev_inv, t2, d_cp);
                mat33_mul(Rprev_inv, nprev_g, n_cp);
                row[0] = n_cp[0]; row[1] = n_cp[1]; row[2] = n_cp[2];
                cross3(s_cp, n_cp, &row[3]);
                float sd[3] = {s_cp[0] - d_cp[0], s_cp[1] - d_cp[1], s_cp[2] - d_cp[2]};
                row[6] = dot3(n_cp, sd);
            }
        }
        store_row8(row, (float)found, &vals[(size_t)i * 8]);
    }
    float h[29];
    reduce29(vals, n, 128, 64, order, h); /* ICPOdometry.cpp:123-124: threads 128, blocks 64 */
    free(vals);
    unpack29(h, A, b, residual);
}

/* ================================================================================================
 * a8  computeRgbResidual -> residualKernel          reduce.cu:668-864
 * The int2 reduction is associative on integers, so summation order is irrelevant.
 * invalid entries keep only .valid = 0 (the reference leaves the other fields uninitialised, A.19).
 * ============================================================================================== */
void kto_rgb_residual(float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* last_depth,
                      const float* next_depth, const uint8_t* last_image, const uint8_t* next_image, int cols, int rows,
                      kto_dataterm* corres_img, float max_depth_delta, const float kt[3], const kto_mat33* krkinv,
                      int* sigma_sum, int* count)
{
    long long cnt = 0, sig = 0;
    const float* K = krkinv->m;
#pragma omp parallel for schedule(static) reduction(+ : cnt, sig)
    for (int k = 0; k < cols * rows; ++k) {
        int i = k / cols, j0 = k - i * cols;
        kto_dataterm corres;
        memset(&corres, 0, sizeof(corres));
        if (j0 < cols - 5 && i < rows - 1) {
            int valid = 1;
            for (int u = imax(i - 2, 0); u < imin(i + 2, rows); u++)
                for (int v = imax(j0 - 2, 0); v < imin(j0 + 2, cols); v++) valid = valid && (next_image[u * cols + v] > 0);
            if (valid) {
                int valx = dIdx[i * cols + j0], valy = dIdy[i * cols + j0];
                float mTwo = (float)((valx * valx) + (valy * valy));
                if (mTwo >= min_scale) {
                    int y = i, x = j0;
                    float d1 = next_depth[y * cols + x];
                    if (!kto_isnan(d1)) {
                        float xf = (float)x, yf = (float)y;
                        float transformed_d1 = fmaf(d1, fmaf(K[6], xf, K[7] * yf) + K[8], kt[2]);
                        int u0 = kto_f2i_rn(fmaf(d1, fmaf(K[0], xf, K[1] * yf) + K[2], kt[0]) / transformed_d1);
                        int v0 = kto_f2i_rn(fmaf(d1, fmaf(K[3], xf, K[4] * yf) + K[5], kt[1]) / transformed_d1);
                        if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
                            float d0 = last_depth[v0 * cols + u0];
                            if (d0 > 0 && fabsf(transformed_d1 - d0) <= max_depth_delta && last_image[v0 * cols + u0] != 0) {
                                corres.zero_x = (int16_t)u0; corres.zero_y = (int16_t)v0;
                                corres.one_x = (int16_t)x; corres.one_y = (int16_t)y;
                                corres.diff = (float)next_image[y * cols + x] - (float)last_image[v0 * cols + u0];
                                corres.valid = 1;
                                cnt += 1;
                                sig += kto_f2i_rz(corres.diff * corres.diff);
                            }
                        }
                    }
                }
            }
        }
        corres_img[k] = corres;
    }
    *count = (int)cnt;
    *sigma_sum = (int)sig; /* int32 wrap-around as on the device */
}

/* ================================================================================================
 * a9  rgbStep -> rgbKernel + reduceSum              reduce.cu:423-607
 * ============================================================================================== */
void kto_rgb_step(const kto_dataterm* corres_img, float sigma, const float* cloud, float fx, float fy,
                  const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows, int order,
                  float A[36], float b[6])
{
    const int n = cols * rows;
    const float flt_eps = 1.19209290E-07F;
    float* vals = malloc((size_t)n * 8 * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const kto_dataterm* c = &corres_img[i];
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        int found = c->valid;
        if (found) {
            float w = sigma + fabsf(c->diff);
            w = w > flt_eps ? 1.0f / w : 1.0f;
            if (sigma == -1) w = 1;
            row[6] = -w * c->diff;
            const float* cp = &cloud[3 * (c->zero_y * cols + c->zero_x)];
            float X = cp[0], Y = cp[1], Z = cp[2];
            float invz = (float)(1.0 / (double)Z);
            float dI_dx_val = w * sobel_scale * (float)dIdx[c->one_y * cols + c->one_x];
            float dI_dy_val = w * sobel_scale * (float)dIdy[c->one_y * cols + c->one_x];
            float v0 = dI_dx_val * fx * invz;
            float v1 = dI_dy_val * fy * invz;
            float v2 = -fmaf(v0, X, v1 * Y) * invz;
            row[0] = v0; row[1] = v1; row[2] = v2;
            /* reduce.cu:478-480.  `-a*b + c*d` is canonicalised to `c*d - a*b` before contraction (LLVM InstCombine), so the
             * fused product of rows 3 and 5 is the second one of the source expression (pinned by oracle/_ref) */
            row[3] = fmaf(Y, v2, -(Z * v1));
            row[4] = fmaf(Z, v0, -(X * v2));
            row[5] = fmaf(X, v1, -(Y * v0));
        }
        store_row8(row, (float)found, &vals[(size_t)i * 8]);
    }
    float h[29];
    reduce29(vals, n, 128, 64, order, h); /* RGBDOdometry.cpp:306-307 */
    free(vals);
    unpack29(h, A, b, NULL);
}

/* ================================================================================================
 * volume helpers                                    device.hpp:61-83
 * ============================================================================================== */
static inline int16_t pack_tsdf(float tsdf)
{
    return (int16_t)imax(-KTO_DIVISOR, imin(KTO_DIVISOR, kto_f2i_rz(tsdf * KTO_DIVISOR)));
}
static inline float unpack_tsdf(int16_t v) { return (float)v / KTO_DIVISOR; }
static inline size_t wrap_index(int x, int y, int z, const int w[3], int N)
{
    return (size_t)((x + w[0]) % N) + (size_t)((y + w[1]) % N) * N + (size_t)((z + w[2]) % N) * N * N;
}

/* initVolume / initColorVolume                      tsdf_volume.cu:56-87, 450-479 */
void kto_init_volume(int16_t* vol, int N) { memset(vol, 0, (size_t)N * N * N * sizeof(int16_t)); }
void kto_init_color_volume(uint8_t* cvol, int N) { memset(cvol, 0, (size_t)N * N * N * 4); }

/* ================================================================================================
 * a11 integrateTsdfVolume -> scaleDepth + tsdf23    tsdf_volume.cu:490-674
 * ============================================================================================== */
void kto_scale_depth(const uint16_t* depth, float* scaled, int cols, int rows, kto_intr intr, int angle_color)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int Dp = depth[y * cols + x];
            float xl = ((float)x - intr.cx) / intr.fx;
            float yl = ((float)y - intr.cy) / intr.fy;
            float lambda = sqrtf(fmaf(xl, xl, yl * yl) + 1);
            if (angle_color) {
                const int ky = 7, kx = 7;
                int ty = imin(y - ky / 2 + ky, rows - 1);
                int tx = imin(x - kx / 2 + kx, cols - 1);
                int count = 0;
                for (int cy = imax(y - ky / 2, 0); cy < ty; ++cy)
                    for (int cx = imax(x - kx / 2, 0); cx < tx; ++cx)
                        if (abs(Dp - (int)depth[cy * cols + cx]) > 200 || Dp == 0) count++;
                if (count > 5) scaled[y * cols + x] = (float)(-Dp) * lambda / 1000.f;
                else scaled[y * cols + x] = (float)Dp * lambda / 1000.f;
            } else
                scaled[y * cols + x] = (float)Dp * lambda / 1000.f;
        }
}

long long kto_integrate_tsdf(const uint16_t* depth_raw, int cols, int rows, kto_intr intr, const float volume_size[3],
                             const kto_mat33* Rcurr_inv, const float tcurr[3], float tranc_dist, int16_t* volume,
                             float* depth_scaled, const int voxel_wrap[3], uint8_t* color_volume,
                             const uint8_t* colors, const float* nmap_curr, int angle_color, int N)
{
    kto_scale_depth(depth_raw, depth_scaled, cols, rows, intr, angle_color);
    const float cell[3] = {volume_size[0] / N, volume_size[1] / N, volume_size[2] / N};
    const float* Ri = Rcurr_inv->m;
    long long updated = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : updated)
    for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
            float v_g_x = fmaf((float)x + 0.5f, cell[0], -tcurr[0]);
            float v_g_y = fmaf((float)y + 0.5f, cell[1], -tcurr[1]);
            float v_g_z = fmaf(0 + 0.5f, cell[2], -tcurr[2]);
            float v_g_part_norm = fmaf(v_g_x, v_g_x, v_g_y * v_g_y);
            float v_x = fmaf(Ri[2], v_g_z, fmaf(Ri[0], v_g_x, Ri[1] * v_g_y)) * intr.fx;
            float v_y = fmaf(Ri[5], v_g_z, fmaf(Ri[3], v_g_x, Ri[4] * v_g_y)) * intr.fy;
            float v_z = fmaf(Ri[8], v_g_z, fmaf(Ri[6], v_g_x, Ri[7] * v_g_y));
            float z_scaled = 0;
            float Rcurr_inv_0_z_scaled = Ri[2] * cell[2] * intr.fx;
            float Rcurr_inv_1_z_scaled = Ri[5] * cell[2] * intr.fy;
            float tranc_dist_inv = 1.0f / tranc_dist;
            /* incremental float walk in z (also on `continue`): quirk A.17 */
            for (int z = 0; z < N; ++z, v_g_z += cell[2], z_scaled += cell[2], v_x += Rcurr_inv_0_z_scaled, v_y += Rcurr_inv_1_z_scaled) {
                float inv_z = 1.0f / fmaf(Ri[8], z_scaled, v_z);
                if (inv_z < 0) continue;
                int coo_x = kto_f2i_rn(fmaf(v_x, inv_z, intr.cx));
                int coo_y = kto_f2i_rn(fmaf(v_y, inv_z, intr.cy));
                if (coo_x >= 0 && coo_y >= 0 && coo_x < cols && coo_y < rows) {
                    float Dp_scaled = depth_scaled[coo_y * cols + coo_x];
                    int no_color = 0;
                    if (Dp_scaled < 0.0) { Dp_scaled = -Dp_scaled; no_color = 1; }
                    float sdf = Dp_scaled - sqrtf(fmaf(v_g_z, v_g_z, v_g_part_norm));
                    if (Dp_scaled != 0 && sdf >= -tranc_dist) {
                        float ncurr_x = nmap_curr[coo_y * cols + coo_x];
                        float ncurr_z = nmap_curr[(coo_y + 2 * rows) * cols + coo_x];
                        if (ncurr_z < 0) ncurr_z = -ncurr_z;
                        float tsdf = fminf(1.0f, sdf * tranc_dist_inv);
                        size_t idx = wrap_index(x, y, z, voxel_wrap, N);
                        float tsdf_prev = unpack_tsdf(volume[idx]);
                        uint8_t* pc = &color_volume[4 * idx];
                        float weight_prev = (float)pc[3]; /* weight lives in colour.w: quirk A.1 */
                        volume[idx] = pack_tsdf(fmaf(tsdf_prev, weight_prev, tsdf) / (weight_prev + 1.0f));
                        pc[3] = f2u8_rz(fminf(weight_prev + 1.0f, KTO_MAX_WEIGHT));
                        ++updated;
                        if ((!kto_isnan(ncurr_x) && !no_color) || (pc[0] == 0 && pc[1] == 0 && pc[2] == 0)) {
                            const float Wrkc = (angle_color ? fminf(1.0f, ncurr_z / KTO_RGB_VIEW_ANGLE_WEIGHT) : 1.0f) * 2.0f;
                            const uint8_t* rgb = &colors[3 * (coo_y * cols + coo_x)];
                            /* (c_prev * W + Wrkc * c_new) / (W + Wrkc), tsdf_volume.cu:627-629: of the two products it is the SECOND that the
                             * compiler fuses into the addition (the first is rounded).  Only a quotient within an ulp of a .5 tie can tell:
                             * one voxel in ~600 random configurations (tests/tools/deep_pin.py seeds 247, 492 pinned it against oracle/_ref). */
                            float new_x = fmaf(Wrkc, (float)rgb[0], (float)pc[0] * weight_prev) / (weight_prev + Wrkc);
                            float new_y = fmaf(Wrkc, (float)rgb[1], (float)pc[1] * weight_prev) / (weight_prev + Wrkc);
                            float new_z = fmaf(Wrkc, (float)rgb[2], (float)pc[2] * weight_prev) / (weight_prev + Wrkc);
                            pc[0] = (uint8_t)imin(255, imax(0, kto_f2i_rn(new_x)));
                            pc[1] = (uint8_t)imin(255, imax(0, kto_f2i_rn(new_y)));
                            pc[2] = (uint8_t)imin(255, imax(0, kto_f2i_rn(new_z)));
                        }
                    }
                }
            }
        }
    return updated;
}

/* ================================================================================================
 * a12 raycast -> rayCastKernel                      ray_caster.cu:56-471
 * ============================================================================================== */
typedef struct {
    const int16_t* volume; const uint8_t* cvol; int N; const int* wrap; float cell[3];
} rc_ctx;
static inline float rc_read_tsdf(const rc_ctx* c, int x, int y, int z) { return unpack_tsdf(c->volume[wrap_index(x, y, z, c->wrap, c->N)]); }
static inline float rc_read_ch(const rc_ctx* c, int x, int y, int z, int ch) { return (float)c->cvol[4 * wrap_index(x, y, z, c->wrap, c->N) + ch]; }
static inline void rc_get_voxel(const rc_ctx* c, const float p[3], int g[3])
{
    g[0] = kto_f2i_rd(p[0] / c->cell[0]);
    g[1] = kto_f2i_rd(p[1] / c->cell[1]);
    g[2] = kto_f2i_rd(p[2] / c->cell[2]);
}
/* shared body of interpolateTrilineary / interpolateColorTrilineary / interpolateHeatTrilineary
 * (ray_caster.cu:160-296). ch < 0: tsdf; ch 0..3: colour channel.  Returns 0 when outside (caller maps). */
static int rc_trilinear(const rc_ctx* c, const float point[3], int ch, float* out)
{
    int g[3];
    rc_get_voxel(c, point, g);
    const int N = c->N;
    if (g[0] <= 0 || g[0] >= N - 1) return 0;
    if (g[1] <= 0 || g[1] >= N - 1) return 0;
    if (g[2] <= 0 || g[2] >= N - 1) return 0;
    float vx = ((float)g[0] + 0.5f) * c->cell[0];
    float vy = ((float)g[1] + 0.5f) * c->cell[1];
    float vz = ((float)g[2] + 0.5f) * c->cell[2];
    g[0] = (point[0] < vx) ? (g[0] - 1) : g[0];
    g[1] = (point[1] < vy) ? (g[1] - 1) : g[1];
    g[2] = (point[2] < vz) ? (g[2] - 1) : g[2];
    float a = fmaf(-((float)g[0] + 0.5f), c->cell[0], point[0]) / c->cell[0];
    float b = fmaf(-((float)g[1] + 0.5f), c->cell[1], point[1]) / c->cell[1];
    float cc = fmaf(-((float)g[2] + 0.5f), c->cell[2], point[2]) / c->cell[2];
    float r[8];
    for (int k = 0; k < 8; ++k) {
        int dx = (k >> 2) & 1, dy = (k >> 1) & 1, dz = k & 1; /* order 000,001,010,011,100,101,110,111 */
        r[k] = ch < 0 ? rc_read_tsdf(c, g[0] + dx, g[1] + dy, g[2] + dz) : rc_read_ch(c, g[0] + dx, g[1] + dy, g[2] + dz, ch);
    }
    float ia = 1 - a, ib = 1 - b, ic = 1 - cc;
    /* sum of 8 products, left-associated, each add contracted with the preceding product's last mul */
    float res = fmaf(r[0] * ia * ib, ic, r[1] * ia * ib * cc);
    res = fmaf(r[2] * ia * b, ic, res);
    res = fmaf(r[3] * ia * b, cc, res);
    res = fmaf(r[4] * a * ib, ic, res);
    res = fmaf(r[5] * a * ib, cc, res);
    res = fmaf(r[6] * a * b, ic, res);
    res = fmaf(r[7] * a * b, cc, res);
    *out = res;
    return 1;
}
static inline float rc_interp_tsdf(const rc_ctx* c, const float p[3])
{
    float r;
    return rc_trilinear(c, p, -1, &r) ? r : kto_nan();
}

long long kto_raycast(kto_intr intr, const kto_mat33* Rcurr, const float tcurr[3], float tranc_dist,
                      const float volume_size[3], const int16_t* volume, float* vmap, float* nmap, int cols, int rows,
                      const int voxel_wrap[3], uint8_t* vmap_curr_color, const uint8_t* color_volume, int N)
{
    rc_ctx c;
    c.volume = volume; c.cvol = color_volume; c.N = N; c.wrap = voxel_wrap;
    c.cell[0] = volume_size[0] / N; c.cell[1] = volume_size[1] / N; c.cell[2] = volume_size[2] / N;
    const float time_step = tranc_dist * 0.8f; /* ray_caster.cu:444 */
    long long steps = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : steps)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            vmap[y * cols + x] = kto_nan();
            nmap[y * cols + x] = kto_nan();
            const float* ray_start = tcurr;
            float rn_[3] = {((float)x - intr.cx) / intr.fx, ((float)y - intr.cy) / intr.fy, 1};
            float ray_next[3];
            mat33_mul(Rcurr, rn_, ray_next);
            ray_next[0] += tcurr[0]; ray_next[1] += tcurr[1]; ray_next[2] += tcurr[2];
            float ray_dir[3] = {ray_next[0] - ray_start[0], ray_next[1] - ray_start[1], ray_next[2] - ray_start[2]};
            normalize3(ray_dir);
            for (int k = 0; k < 3; ++k) ray_dir[k] = (ray_dir[k] == 0.f) ? (float)1e-15 : ray_dir[k];
            /* getMinTime / getMaxTime  ray_caster.cu:56-74 */
            float tmin[3], tmax[3];
            for (int k = 0; k < 3; ++k) {
                tmin[k] = ((ray_dir[k] > 0 ? 0.f : volume_size[k]) - ray_start[k]) / ray_dir[k];
                tmax[k] = ((ray_dir[k] > 0 ? volume_size[k] : 0.f) - ray_start[k]) / ray_dir[k];
            }
            float time_start_volume = fmaxf(fmaxf(tmin[0], tmin[1]), tmin[2]);
            float time_exit_volume = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
            time_start_volume = fmaxf(time_start_volume, 0.f);
            if (time_start_volume >= time_exit_volume) continue;
            float time_curr = time_start_volume;
            float p[3];
            int g[3];
            for (int k = 0; k < 3; ++k) p[k] = fmaf(ray_dir[k], time_curr, ray_start[k]);
            rc_get_voxel(&c, p, g);
            g[0] = imax(0, imin(g[0], N - 1)); g[1] = imax(0, imin(g[1], N - 1)); g[2] = imax(0, imin(g[2], N - 1));
            float tsdf = rc_read_tsdf(&c, g[0], g[1], g[2]);
            const float max_time = 3 * (volume_size[0] + volume_size[1] + volume_size[2]);
            for (; time_curr < max_time; time_curr += time_step) {
                float tsdf_prev = tsdf;
                float tn = time_curr + time_step;
                for (int k = 0; k < 3; ++k) p[k] = fmaf(ray_dir[k], tn, ray_start[k]);
                rc_get_voxel(&c, p, g);
                /* checkInds: z tested against VOLUME_X (quirk A.13; all sides equal here) */
                if (!(g[0] >= 0 && g[1] >= 0 && g[2] >= 0 && g[0] < N && g[1] < N && g[2] < N)) break;
                tsdf = rc_read_tsdf(&c, g[0], g[1], g[2]);
                ++steps;
                if (tsdf_prev < 0.f && tsdf > 0.f) break;
                if (tsdf_prev > 0.f && tsdf < 0.f) {
                    float Ftdt = rc_interp_tsdf(&c, p);
                    if (kto_isnan(Ftdt)) break;
                    float p0[3];
                    for (int k = 0; k < 3; ++k) p0[k] = fmaf(ray_dir[k], time_curr, ray_start[k]);
                    float Ft = rc_interp_tsdf(&c, p0);
                    if (kto_isnan(Ft)) break;
                    float Ts = time_curr - time_step * Ft / (Ftdt - Ft);
                    float vf[3];
                    for (int k = 0; k < 3; ++k) vf[k] = fmaf(ray_dir[k], Ts, ray_start[k]);
                    vmap[y * cols + x] = vf[0];
                    vmap[(y + rows) * cols + x] = vf[1];
                    vmap[(y + 2 * rows) * cols + x] = vf[2];
                    int gg[3];
                    rc_get_voxel(&c, p0, gg);
                    /* colour + heat: float -> uchar truncation (quirk A.20); outside -> black / NaN -> 0 */
                    uint8_t* pc = &vmap_curr_color[4 * (y * cols + x)];
                    float col;
                    for (int ch = 0; ch < 3; ++ch) pc[ch] = rc_trilinear(&c, vf, ch, &col) ? f2u8_rz(col) : 0;
                    pc[3] = rc_trilinear(&c, vf, 3, &col) ? f2u8_rz(col) : 0;
                    if (gg[0] > 1 && gg[1] > 1 && gg[2] > 1 && gg[0] < N - 2 && gg[1] < N - 2 && gg[2] < N - 2) {
                        float n[3], t[3];
                        for (int k = 0; k < 3; ++k) {
                            t[0] = vf[0]; t[1] = vf[1]; t[2] = vf[2];
                            t[k] += c.cell[k];
                            float F1 = rc_interp_tsdf(&c, t);
                            t[0] = vf[0]; t[1] = vf[1]; t[2] = vf[2];
                            t[k] -= c.cell[k];
                            float F2 = rc_interp_tsdf(&c, t);
                            n[k] = F1 - F2;
                        }
                        normalize3(n);
                        nmap[y * cols + x] = n[0];
                        nmap[(y + rows) * cols + x] = n[1];
                        nmap[(y + 2 * rows) * cols + x] = n[2];
                    }
                    break;
                }
            }
        }
    return steps;
}

/* ================================================================================================
 * a14 clearVolume{X,Y,Z}{,Back}{,c}                 tsdf_volume.cu:88-448
 * The launch geometry of the X variants is emulated (quirk A.15).
 * ============================================================================================== */
static inline void clear_voxel(void* vol, int elem_size, size_t idx)
{
    if (elem_size == 2) ((int16_t*)vol)[idx] = 0;
    else ((uint32_t*)vol)[idx] = 0;
}
static int wrap_base(int currentVoxelWrap, int N)
{
    return currentVoxelWrap > 0 ? currentVoxelWrap % N : N - ((-currentVoxelWrap) % N);
}
void kto_clear_volume(void* vol, int elem_size, int N, int axis, int back, int currentVoxelWrap, int deltaVoxelWrap)
{
    if (axis == 0) {
        /* clearVolumeX / clearVolumeXBack :117-237, kernel clearVolumeInX :88-115 */
        int remainder = (deltaVoxelWrap - currentVoxelWrap) % 16;
        if (remainder != 0) remainder = (deltaVoxelWrap - currentVoxelWrap) + 16 - remainder;
        else remainder = abs(deltaVoxelWrap - currentVoxelWrap);
        int grid_x = (remainder + 15) / 16; /* divUp(remainder, 16); negative remainder -> <= 0 blocks */
        int bottom, num;
        if (!back) {
            bottom = wrap_base(currentVoxelWrap, N);
            num = -(currentVoxelWrap - deltaVoxelWrap);
        } else {
            int base = wrap_base(currentVoxelWrap, N);
            int top = (base + N) % N;
            num = currentVoxelWrap - deltaVoxelWrap;
            bottom = top - num;
            if (bottom < 0) bottom = N + bottom;
        }
        int nthreads_x = grid_x * 16;
        bottom %= N;
        const int cachedWrap = (bottom + num) % N;
        const int wrap = cachedWrap != bottom + num;
        for (int tx = 0; tx < nthreads_x; ++tx) {
            int x = (tx + bottom) % N;
            if (!wrap ? (x >= bottom && x <= cachedWrap) : (x >= bottom || x <= cachedWrap))
                for (int y = 0; y < N; ++y)
                    for (int z = 0; z < N; ++z) clear_voxel(vol, elem_size, (size_t)x + (size_t)y * N + (size_t)z * N * N);
        }
        return;
    }
    /* Y and Z variants: N x N threads each walking the slab :239-448 (thread y is the z index in the Y kernels) */
    for (int ty = 0; ty < N; ++ty)
        for (int tx = 0; tx < N; ++tx) {
            if (!back) {
                int bottom = wrap_base(currentVoxelWrap, N);
                int numUp = -(currentVoxelWrap - deltaVoxelWrap);
                while (numUp >= 0) {
                    size_t idx = axis == 1 ? (size_t)tx + (size_t)(bottom++ % N) * N + (size_t)ty * N * N
                                           : (size_t)tx + (size_t)ty * N + (size_t)(bottom++ % N) * N * N;
                    clear_voxel(vol, elem_size, idx);
                    numUp--;
                }
            } else {
                int base = wrap_base(currentVoxelWrap, N);
                int top = (base + N) % N;
                int numDown = currentVoxelWrap - deltaVoxelWrap;
                while (numDown >= 0) {
                    size_t idx = axis == 1 ? (size_t)tx + (size_t)(top-- % N) * N + (size_t)ty * N * N
                                           : (size_t)tx + (size_t)ty * N + (size_t)(top-- % N) * N * N;
                    clear_voxel(vol, elem_size, idx);
                    if (top < 0) top = N - 1;
                    numDown--;
                }
            }
        }
}

/* ================================================================================================
 * a15 extractCloudSlice -> extractKernelSlice       extract.cu:79-419
 * Output order here is z-major then y, x (the reference's order is nondeterministic: atomicAdd
 * compaction, extract.cu:255); compare as sorted sets.
 * ============================================================================================== */
typedef struct { const int16_t* vol; const uint8_t* cvol; const int* wrap; int N; } ex_ctx;
static inline float ex_fetch(const ex_ctx* e, int x, int y, int z, int* weight)
{
    size_t i = wrap_index(x, y, z, e->wrap, e->N);
    *weight = e->cvol[4 * i + 3];
    return unpack_tsdf(e->vol[i]);
}
size_t kto_extract_cloud_slice(const int16_t* volume, const float volume_size[3], kto_point* out, size_t out_cap,
                               const int voxel_wrap[3], const uint8_t* color_volume, int minX, int maxX, int minY,
                               int maxY, int minZ, int maxZ, int subsample, const int real_voxel_wrap[3], int N)
{
    ex_ctx e = {volume, color_volume, voxel_wrap, N};
    const float cell[3] = {volume_size[0] / N, volume_size[1] / N, volume_size[2] / N};
    size_t count = 0;
    for (int z = minZ; z < maxZ; z += subsample)
        for (int y = imax(minY, 0); y < imin(maxY, N); ++y)
            for (int x = imax(minX, 0); x < imin(maxX, N); ++x) {
                if (x % subsample != 0 || y % subsample != 0) continue;
                int W;
                float F = ex_fetch(&e, x, y, z, &W);
                if (!(W != 0 && F != 1.f)) continue;
                float V[3] = {((float)x + 0.5f) * cell[0], ((float)y + 0.5f) * cell[1], ((float)z + 0.5f) * cell[2]};
                for (int axis = 0; axis < 3; ++axis) {
                    int nx = x + (axis == 0), ny = y + (axis == 1), nz = z + (axis == 2);
                    if (axis == 0 && !(x + 1 < N)) continue;
                    if (axis == 1 && !(y + 1 < N)) continue;
                    /* z + 1 at z == N-1 wraps modulo N through the storage index (quirk) */
                    int Wn;
                    float Fn = ex_fetch(&e, nx, ny, nz, &Wn);
                    if (!(Wn != 0 && Fn != 1.f)) continue;
                    if (!((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) continue;
                    float p[3] = {V[0], V[1], V[2]};
                    float Vn = V[axis] + cell[axis];
                    float d_inv = 1.f / (fabsf(F) + fabsf(Fn));
                    /* (V * |Fn| + Vn * |F|) * d_inv, extract.cu:166, :195, :224.  For dx and dy the left product is the one
                     * contracted; in the dz block V.z and Vnz are defined inside the z loop and LLVM's canonical operand
                     * order puts Vnz * |F| first, so there the other product is fused (pinned by oracle/_ref). */
                    p[axis] = (axis == 2 ? fmaf(Vn, fabsf(F), V[axis] * fabsf(Fn)) : fmaf(V[axis], fabsf(Fn), Vn * fabsf(F))) * d_inv;
                    size_t ni = wrap_index(nx, ny, nz, voxel_wrap, N);
                    if (count < out_cap) {
                        kto_point* o = &out[count];
                        memset(o, 0, sizeof(*o));
                        /* store_point_type extract.cu:307-317: x + realVoxelWrap.x * cell_size.x - half.  The product is invariant over
                         * the whole kernel, so the compiler computes it once in front of the loops, where it cannot fuse with the
                         * addition inside them: NOT contracted (pinned by oracle/_ref with real wraps of a few hundred voxels,
                         * tests/test_oracle_vs_ref.py::test_randomized_sweep; with small wraps both forms round alike). */
                        o->x = (p[0] + (float)real_voxel_wrap[0] * cell[0]) - ((cell[0] * N) / 2);
                        o->y = (p[1] + (float)real_voxel_wrap[1] * cell[1]) - ((cell[1] * N) / 2);
                        o->z = (p[2] + (float)real_voxel_wrap[2] * cell[2]) - ((cell[2] * N) / 2);
                        /* r = colour.x of the NEIGHBOUR, stored swapped: ptr->r = b, ptr->b = r (quirk A.14) */
                        o->r = color_volume[4 * ni + 2];
                        o->g = color_volume[4 * ni + 1];
                        o->b = color_volume[4 * ni + 0];
                        o->a = (uint8_t)W;
                    }
                    ++count;
                }
            }
    return count < out_cap ? count : out_cap;
}

/* ================================================================================================
 * f2  CloudSliceProcessor::process, the per-slice stage      backend/CloudSliceProcessor.cpp:87-163
 *   weight cull -> pcl::VoxelGrid<PointXYZRGB> (leaf = voxel size) -> pcl::NormalEstimation (kNN 20) -> PointXYZRGBNormal
 * PCL 1.7 (README.md:14-31) is not vendored with the reference and not installed here: this restates its published
 * algorithms -- filters/impl/voxel_grid.hpp (applyFilter), common/impl/centroid.hpp (computeMeanAndCovarianceMatrix, float
 * single-pass form), features/normal_3d.h (solvePlaneParameters, flipNormalTowardsViewpoint with the default sensor origin 0),
 * common/impl/eigen.hpp (computeRoots, eigen33) -- in float without contraction (the reference builds host code with -msse3).
 * PARITY UNPINNED against PCL itself.  Two orders PCL leaves to its implementation are FIXED here and in the HIP stage:
 *   - VoxelGrid sorts (leaf index, point) pairs with std::sort, which is not stable: here the points of a leaf are summed in
 *     their original order;
 *   - FLANN breaks distance ties arbitrarily: here neighbours are ordered by (squared distance, index).
 * ============================================================================================== */
typedef struct { unsigned key; unsigned src; } sp_pair;
static int sp_pair_cmp(const void* a, const void* b)
{
    const sp_pair *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->src < y->src ? -1 : (x->src > y->src);
}
/* sin / cos / atan2 of pcl::computeRoots (common/impl/eigen.hpp), restated: the C library's and the device library's versions differ
 * from each other in the last bits, and a normal is only as reproducible as they are.  The stage needs them on a small domain --
 * atan2(y >= 0, x) in [0, pi], then sin / cos of a third of that, in [0, pi / 3] -- so both sides (this file and csrc/kt_slice.hip,
 * the same operations in the same order, every fused multiply-add written out) use the Cephes single-precision kernels:
 * atanf on [0, tan(pi / 8)] with one reduction step, sinf / cosf on [0, pi / 4] with the complement for the rest.  Absolute error
 * ~1e-7, like the libraries'; PCL itself (libm) is not reproduced to the bit by either -- f2 is "parity unpinned" by necessity. */
static float sp_atan01(float a)   /* atan(a), 0 <= a <= 1 */
{
    float y0 = 0.0f, x = a;
    if (a > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (a - 1.0f) / (a + 1.0f); }   /* tan(pi / 8) */
    const float z = x * x;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, x, x);
}
static float sp_atan2_pos(float y, float x)   /* atan2f(y, x) for y >= 0 */
{
    const float ax = fabsf(x);
    const float hi = ax > y ? ax : y, lo = ax > y ? y : ax;
    if (!(hi > 0.0f)) return 0.0f;                       /* atan2(0, 0) = 0; atan2(0, x < 0) = pi is the lo == 0, x < 0 case below */
    float r = sp_atan01(lo / hi);
    if (y > ax) r = 1.57079632679489661923f - r;
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return r;
}
static void sp_sincos(float t, float* s, float* c)   /* 0 <= t <= pi / 3 (+ a few ulp) */
{
    const int swap = t > 0.78539816339744830962f;
    const float x = swap ? 1.57079632679489661923f - t : t;
    const float z = x * x;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sn = fmaf(ps * z, x, x);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cs = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    *s = swap ? cs : sn;
    *c = swap ? sn : cs;
}
/* test access (tests/test_oracle_kat.py pins the three against libm: a wrong coefficient or reduction would otherwise pass on both sides,
 * HIP and oracle sharing this restatement) */
float kto_test_sp_atan2_pos(float y, float x) { return sp_atan2_pos(y, x); }
void kto_test_sp_sincos(float t, float* s, float* c) { sp_sincos(t, s, c); }
static void sp_compute_roots2(float b, float c, float r[3])
{
    r[0] = 0.f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    const float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}
static void sp_compute_roots(const float m[9], float r[3])
{
    const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < 1.1920929e-07f) { sp_compute_roots2(c2, c1, r); return; }
    const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = sqrtf(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = sqrtf(-a_over_3);
    const float theta = sp_atan2_pos(sqrtf(-q), half_b) * s_inv3;
    float cos_theta, sin_theta;
    sp_sincos(theta, &sin_theta, &cos_theta);
    r[0] = c2_over_3 + 2.0f * rho * cos_theta;
    r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    }
    if (r[0] <= 0.0f) sp_compute_roots2(c2, c1, r);
}
/* pcl::eigen33(mat, eigenvalue, eigenvector): smallest eigenvalue and its eigenvector */
static void sp_eigen33(const float mat[9], float* eigenvalue, float v[3])
{
    float scale = 0.0f;
    for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(mat[i]));
    if (scale <= 1.17549435e-38f) scale = 1.0f;
    float s[9], roots[3];
    for (int i = 0; i < 9; ++i) s[i] = mat[i] / scale;
    sp_compute_roots(s, roots);
    *eigenvalue = roots[0] * scale;
    s[0] -= roots[0]; s[4] -= roots[0]; s[8] -= roots[0];
    const float *r0 = &s[0], *r1 = &s[3], *r2 = &s[6];
    const float v1[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    const float v2[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
    const float v3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const float l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2], l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2],
                l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
    const float* w = v3;
    float len = l3;
    if (l1 >= l2 && l1 >= l3) { w = v1; len = l1; }
    else if (l2 >= l1 && l2 >= l3) { w = v2; len = l2; }
    const float sl = sqrtf(len);
    v[0] = w[0] / sl; v[1] = w[1] / sl; v[2] = w[2] / sl;
}

/* ================================================================================================
 * f1  CloudSliceProcessor::save: the final pcl::VoxelGrid<pcl::PointXYZRGBNormal> over the concatenated processed clouds
 * (backend/CloudSliceProcessor.cpp:197-218, run when extractOverlap && !saveOverlap) and pcl::io::savePCDFile(file, cloud, true)
 * (:224-226).  PCL 1.7 filters/impl/voxel_grid.hpp with downsample_all_data_ (the default): every registered field of the point
 * (x y z rgb normal_x normal_y normal_z curvature, 8 floats) is averaged as a float -- normals are NOT renormalised -- plus r, g, b
 * as three more floats (the "RGB special case"); the averaged `rgb` float is then overwritten by the packed
 * (int(r) << 16 | int(g) << 8 | int(b)) (alpha byte 0).  Leaves in key order; points of a leaf summed in input order (std::sort is
 * unstable: fixed as in kto_slice_process); `centroid /= n` is Eigen 3.2's multiplication by 1 / n.  The output points are value-
 * initialised PointXYZRGBNormal (data[3] = 1, data_n[3] = 0, the two floats behind curvature 0).  PARITY UNPINNED against PCL itself.
 * in / out: 48-byte points {x y z 1 | nx ny nz 0 | bgra, curvature, 0, 0}; returns the count (<= n).
 * ============================================================================================== */
size_t kto_voxel_grid_normal(const float* in48, size_t n, float leaf, float* out48)
{
    if (n == 0) return 0;
    const float inv_leaf = 1.0f / leaf;
    float mn[3] = {in48[0], in48[1], in48[2]}, mx[3] = {in48[0], in48[1], in48[2]};
    for (size_t i = 1; i < n; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], in48[12 * i + a]); mx[a] = fmaxf(mx[a], in48[12 * i + a]); }
    /* "Leaf size is too small for the input dataset": dx * dy * dz of the BOUNDING BOX in leaves must fit an int32 */
    const long long dx = (long long)((mx[0] - mn[0]) * inv_leaf) + 1, dy = (long long)((mx[1] - mn[1]) * inv_leaf) + 1,
                    dz = (long long)((mx[2] - mn[2]) * inv_leaf) + 1;
    if (dx * dy * dz > 2147483647LL) { memcpy(out48, in48, n * 48); return n; }
    int min_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)floorf(mn[a] * inv_leaf);
        div_b[a] = (int)floorf(mx[a] * inv_leaf) - min_b[a] + 1;
    }
    sp_pair* pr = malloc(n * sizeof(sp_pair));
    for (size_t i = 0; i < n; ++i) {
        const float* p = &in48[12 * i];
        const int i0 = (int)(floorf(p[0] * inv_leaf) - (float)min_b[0]), i1 = (int)(floorf(p[1] * inv_leaf) - (float)min_b[1]),
                  i2 = (int)(floorf(p[2] * inv_leaf) - (float)min_b[2]);
        pr[i].key = (unsigned)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]);
        pr[i].src = (unsigned)i;
    }
    qsort(pr, n, sizeof(sp_pair), sp_pair_cmp);
    size_t nout = 0, i = 0;
    while (i < n) {
        size_t j = i;
        float c[11];
        for (; j < n && pr[j].key == pr[i].key; ++j) {
            const float* p = &in48[12 * pr[j].src];
            const unsigned char* bgra = (const unsigned char*)&p[8];
            /* field order of POINT_CLOUD_REGISTER_POINT_STRUCT(PointXYZRGBNormal): x y z rgb normal_x normal_y normal_z curvature */
            const float t[11] = {p[0], p[1], p[2], p[8], p[4], p[5], p[6], p[9], (float)bgra[2], (float)bgra[1], (float)bgra[0]};
            if (j == i) memcpy(c, t, sizeof(c));
            else for (int a = 0; a < 11; ++a) c[a] += t[a];
        }
        const float inv_cnt = 1.0f / (float)(j - i);
        for (int a = 0; a < 11; ++a) c[a] *= inv_cnt;
        float* o = &out48[12 * nout];
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = 1.0f;
        o[4] = c[4]; o[5] = c[5]; o[6] = c[6]; o[7] = 0.0f;
        const int rgb = ((int)c[8] << 16) | ((int)c[9] << 8) | (int)c[10];
        memcpy(&o[8], &rgb, 4);
        o[9] = c[7]; o[10] = 0.0f; o[11] = 0.0f;
        ++nout;
        i = j;
    }
    free(pr);
    return nout;
}

/* pcl::io::savePCDFile(file, cloud, true) = PCDWriter::writeBinary<PointXYZRGBNormal> (io/impl/pcd_io.hpp, PCL 1.7): the header
 * generateHeader writes, "DATA binary", then per point the registered fields back to back (32 bytes: x y z rgb normal_x normal_y
 * normal_z curvature; the struct's padding floats are not fields).  out must hold 512 + 32 n bytes; returns the byte count. */
size_t kto_pcd_binary(const float* pts48, size_t n, unsigned char* out)
{
    const int h = sprintf((char*)out,
                          "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb normal_x normal_y normal_z curvature\n"
                          "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                          "POINTS %zu\nDATA binary\n", n, n);
    unsigned char* o = out + h;
    static const int field_at[8] = {0, 1, 2, 8, 4, 5, 6, 9};
    for (size_t i = 0; i < n; ++i)
        for (int f = 0; f < 8; ++f, o += 4) memcpy(o, &pts48[12 * i + field_at[f]], 4);
    return (size_t)(o - out);
}

/* in: n points (32 B, kto_point); out: up to n points of 48 B {x y z 1 | nx ny nz 0 | b g r a, curvature, 0, 0}.  Returns the count. */
size_t kto_slice_process(const kto_point* in, size_t n, int weight_cull, float leaf, int k, float* out48)
{
    kto_point* pts = malloc((n ? n : 1) * sizeof(kto_point));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) /* CloudSliceProcessor.cpp:99-117 */
        if (!(weight_cull > 0) || in[i].a >= weight_cull) pts[m++] = in[i];
    if (m == 0) { free(pts); return 0; }
    /* ---- VoxelGrid::applyFilter ---- */
    const float inv_leaf = 1.0f / leaf;
    float mn[3] = {pts[0].x, pts[0].y, pts[0].z}, mx[3] = {pts[0].x, pts[0].y, pts[0].z};
    for (size_t i = 1; i < m; ++i) {
        const float p[3] = {pts[i].x, pts[i].y, pts[i].z};
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], p[a]); mx[a] = fmaxf(mx[a], p[a]); }
    }
    int min_b[3], max_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)floorf(mn[a] * inv_leaf);
        max_b[a] = (int)floorf(mx[a] * inv_leaf);
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    size_t nout = 0;
    int passthrough = 0;
    float* cen = NULL;        /* per leaf: x y z r g b */
    int* cell = NULL;         /* per leaf: i j k */
    /* voxel_grid.hpp: dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inverse_leaf_size_[0]) + 1, ...; dx * dy * dz > INT32_MAX */
    const long long cells = ((long long)((mx[0] - mn[0]) * inv_leaf) + 1) * ((long long)((mx[1] - mn[1]) * inv_leaf) + 1) * ((long long)((mx[2] - mn[2]) * inv_leaf) + 1);
    if (cells > 2147483647LL) { /* "Leaf size is too small for the input dataset": the cloud passes through unfiltered */
        cen = malloc(m * 6 * sizeof(float)); cell = malloc(m * 3 * sizeof(int));
        for (size_t i = 0; i < m; ++i) {
            cen[6 * i] = pts[i].x; cen[6 * i + 1] = pts[i].y; cen[6 * i + 2] = pts[i].z;
            cen[6 * i + 3] = pts[i].r; cen[6 * i + 4] = pts[i].g; cen[6 * i + 5] = pts[i].b;
            cell[3 * i] = cell[3 * i + 1] = cell[3 * i + 2] = 0;
        }
        nout = m;
        passthrough = 1;
    } else {
        sp_pair* pr = malloc(m * sizeof(sp_pair));
        for (size_t i = 0; i < m; ++i) {
            const int i0 = (int)(floorf(pts[i].x * inv_leaf) - (float)min_b[0]), i1 = (int)(floorf(pts[i].y * inv_leaf) - (float)min_b[1]),
                      i2 = (int)(floorf(pts[i].z * inv_leaf) - (float)min_b[2]);
            pr[i].key = (unsigned)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]);
            pr[i].src = (unsigned)i;
        }
        qsort(pr, m, sizeof(sp_pair), sp_pair_cmp);
        cen = malloc(m * 6 * sizeof(float)); cell = malloc(m * 3 * sizeof(int));
        size_t i = 0;
        while (i < m) {
            size_t j = i;
            float acc[6] = {0, 0, 0, 0, 0, 0};
            while (j < m && pr[j].key == pr[i].key) {
                const kto_point* p = &pts[pr[j].src];
                acc[0] += p->x; acc[1] += p->y; acc[2] += p->z; acc[3] += (float)p->r; acc[4] += (float)p->g; acc[5] += (float)p->b;
                ++j;
            }
            /* `centroid /= static_cast<float>(n)` on an Eigen::VectorXf: Eigen 3.2 (README.md:14-31: the Ubuntu 14.04 / 15.04 packages)
             * evaluates a floating-point `/= s` as `*= Scalar(1) / s` (Core/SelfCwiseBinaryOp.h; true division only from Eigen 3.3 on) */
            const float inv_cnt = 1.0f / (float)(j - i);
            for (int a = 0; a < 6; ++a) cen[6 * nout + a] = acc[a] * inv_cnt;
            const unsigned key = pr[i].key;
            cell[3 * nout] = (int)(key % (unsigned)div_b[0]);
            cell[3 * nout + 1] = (int)((key / (unsigned)div_b[0]) % (unsigned)div_b[1]);
            cell[3 * nout + 2] = (int)(key / ((unsigned)div_b[0] * (unsigned)div_b[1]));
            ++nout;
            i = j;
        }
        free(pr);
    }
    /* ---- NormalEstimation::computeFeature ---- */
    int kk = (size_t)k < nout ? k : (int)nout;
    if (kk > 64) kk = 64;   /* the neighbour list below holds 64 (the reference asks for 20) */
    if (kk < 1) kk = 1;
#pragma omp parallel for schedule(static)
    for (long long q = 0; q < (long long)nout; ++q) {
        const float px = cen[6 * q], py = cen[6 * q + 1], pz = cen[6 * q + 2];
        float bd[64]; int bi[64]; int cnt = 0;     /* the k nearest so far, ascending (distance, index) */
        for (size_t j = 0; j < nout; ++j) {
            const float dx = cen[6 * j] - px, dy = cen[6 * j + 1] - py, dz = cen[6 * j + 2] - pz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (cnt == kk && !(d < bd[cnt - 1])) continue;        /* an equal distance with a larger index never displaces */
            int pos = cnt < kk ? cnt : kk - 1;
            while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
            bd[pos] = d; bi[pos] = (int)j;
            if (cnt < kk) ++cnt;
        }
        float* o = &out48[12 * q];
        o[0] = px; o[1] = py; o[2] = pz; o[3] = 1.0f;
        o[7] = 0.0f; o[10] = 0.0f; o[11] = 0.0f;
        /* VoxelGrid: r, g, b = (uint8) of the float means, packed into rgb with a zero alpha byte (`output = *input_` of the
         * leaf-too-small case keeps the point, its weight byte included) */
        unsigned char* c = (unsigned char*)&o[8];
        c[0] = (unsigned char)cen[6 * q + 5]; c[1] = (unsigned char)cen[6 * q + 4]; c[2] = (unsigned char)cen[6 * q + 3];
        c[3] = passthrough ? pts[q].a : 0;
        if (cnt < 3) { o[4] = o[5] = o[6] = o[9] = NAN; continue; }
        float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < cnt; ++t) {
            const float x = cen[6 * bi[t]], y = cen[6 * bi[t] + 1], z = cen[6 * bi[t] + 2];
            acc[0] += x * x; acc[1] += x * y; acc[2] += x * z; acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
            acc[6] += x; acc[7] += y; acc[8] += z;
        }
        const float inv_cnt = 1.0f / (float)cnt;   /* `accu /= static_cast<Scalar>(point_count)` (centroid.hpp), Eigen 3.2: see above */
        for (int a = 0; a < 9; ++a) acc[a] *= inv_cnt;
        float cov[9];
        cov[0] = acc[0] - acc[6] * acc[6]; cov[1] = acc[1] - acc[6] * acc[7]; cov[2] = acc[2] - acc[6] * acc[8];
        cov[4] = acc[3] - acc[7] * acc[7]; cov[5] = acc[4] - acc[7] * acc[8]; cov[8] = acc[5] - acc[8] * acc[8];
        cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
        float ev, nv[3];
        sp_eigen33(cov, &ev, nv);
        const float eig_sum = cov[0] + cov[4] + cov[8];
        o[9] = eig_sum != 0 ? fabsf(ev / eig_sum) : 0;
        /* flipNormalTowardsViewpoint, viewpoint = sensor origin (0, 0, 0) */
        const float cos_theta = ((0.0f - px) * nv[0] + (0.0f - py) * nv[1]) + (0.0f - pz) * nv[2];
        if (cos_theta < 0) { nv[0] *= -1; nv[1] *= -1; nv[2] *= -1; }
        o[4] = nv[0]; o[5] = nv[1]; o[6] = nv[2];
    }
    free(cen); free(cell); free(pts);
    return nout;
}
