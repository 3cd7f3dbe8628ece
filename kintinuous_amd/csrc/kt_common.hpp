// kt_common.hpp -- shared host/device helpers for libkt_hip.so (gfx950 only).
//
// Arithmetic contract (DESIGN.md "Arithmetic"; the test oracle restates the same rules independently): IEEE binary32,
// correctly rounded '/' and sqrtf (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), built with
// -ffp-contract=off so a*b+c fuses ONLY where __builtin_fmaf is written; rsqrtf -> 1/sqrtf;
// __expf -> kt_expf (explicit fmaf chain); CUDA __float2int_r{n,z,d} semantics (saturate, NaN -> 0)
// come for free from v_cvt_i32_f32.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kt_abi.h"
#include "kt_debug.h"   // test / analysis hooks (not part of the boundary)

#define KT_DIVISOR 32767                 // internal.h:237
#define KT_RGB_VIEW_ANGLE_WEIGHT 0.75f   // internal.h:241
#define KT_MAX_WEIGHT 128.0f             // tsdf_volume.cu:481-488
#define KT_LEVELS 4                      // ICPOdometry.h:52

struct kt_integrate_scratch;  // kt_volume.hip

struct kt_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    // scratch for reductions: hand-off granules (two sets of u64[15][256] {sum 2p, sum 2p + 1}, all ones = empty; kt_track.hip) + final (float[32])
    double* red_partials;
    float* red_out;        // device, 32 floats
    float* red_out_host;   // pinned host mirror
    unsigned int* counters;  // device: [0] ticket, [1] extract global count, [2..] spare
    int* int_out_host;       // pinned, small
    int red_max_blocks;
    kt_integrate_scratch* integ;   // integrate scratch (pixel records, z tables, intervals, task list), created on first use
    float* bil_lut;                // bilateral tap weights [27][396] (kt_image.hip), built on first use
    int fault_skip, fault_count;   // test hook kt_debug_handoff_fault: after fault_skip more ICP reduction launches, fault_count launches lose a publisher
    unsigned long long* pose_gran;   // device, 16 x {float, seq}: the in-kernel pose hand-over of kt_icp_level_kernel (kt_track.hip)
    unsigned int odo_seq;            // its sequence counter
    unsigned int red_epoch;  // launch counter; the tag of the host-form residual launch's granules (kt_track.hip)
    void* track_state;       // device kt_track_state of kt_icp_track (kt_track.hip), created on first use
    void* slice_ws;          // kt_slice_ws of the host-array kt_slice_process (kt_slice.hip), created on first use
};

void kt_set_error(const char* fmt, ...);
void kt_integrate_scratch_free(kt_ctx* c);
int kt_check(hipError_t e, const char* what, const char* file, int line);
#define KT_HIP(expr)                                                        \
    do {                                                                    \
        int _s = kt_check((expr), #expr, __FILE__, __LINE__);               \
        if (_s != KT_OK) return _s;                                         \
    } while (0)
#define KT_LAUNCH_CHECK() KT_HIP(hipGetLastError())
#define KT_TRY(expr) do { int _s = (expr); if (_s != KT_OK) return _s; } while (0)
#define KT_ARG(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { kt_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); return KT_ERR_ARG; } \
    } while (0)

static inline int kt_div_up(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers ------------------------------------------------------------------------------
#ifdef __HIPCC__
struct f3 { float x, y, z; };

__device__ __forceinline__ float kt_nan() { return __int_as_float(0x7fffffff); }  // limits.hpp quiet_NaN
__device__ __forceinline__ bool kt_isnan(float x) { return x != x; }

// a * b + c and a * b with a, b < 2^24 (low 32 bits of the 48-bit product): v_mad_u32_u24 / v_mul_u32_u24 run at full
// rate, a 32 x 32 multiply (v_mul_lo_u32, v_mad_u64_u32) at a quarter of it.  As inline assembly: given the mul24 intrinsic LLVM proves
// the operands small and turns it back into a generic 32-bit multiply, which it then selects as v_mad_u64_u32 / v_mul_lo_u32.
__device__ __forceinline__ unsigned int kt_mad24(unsigned int a, unsigned int b, unsigned int c)
{
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// ... with a wave-uniform b (taken from its SGPR)
__device__ __forceinline__ unsigned int kt_mad24u(unsigned int a, unsigned int b, unsigned int c)
{
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned int kt_mul24(unsigned int a, unsigned int b)
{
    unsigned int r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// v_cvt_i32_f32: truncates, saturates, NaN -> 0  (== CUDA cvt.r*i.s32.f32 after the rounding step)
__device__ __forceinline__ int kt_cvt_i32(float x)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int kt_f2i_rn(float x) { return kt_cvt_i32(__builtin_rintf(x)); }
__device__ __forceinline__ int kt_f2i_rz(float x) { return kt_cvt_i32(x); }
__device__ __forceinline__ int kt_f2i_rd(float x) { return kt_cvt_i32(__builtin_floorf(x)); }
// float -> uchar, cvt.rzi.u8.f32: truncate, clamp [0,255], NaN -> 0
__device__ __forceinline__ unsigned char kt_f2u8_rz(float x)
{
    int v = kt_cvt_i32(x);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    return (unsigned char)v;
}
__device__ __forceinline__ short kt_f2s16_rz(float x)
{
    int v = kt_cvt_i32(x);
    v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
    return (short)v;
}

// restatement of __expf (bilateral_pyrdown.cu:89); same steps as the oracle's kto_expf
__device__ __forceinline__ float kt_expf(float x)
{
    float t = x * 1.44269504088896341f;
    if (!(t >= -125.0f)) return 0.0f;
    if (t >= 128.0f) return __builtin_inff();   // ex2.approx overflows to +inf (reached when bilateralKernel's int product wraps)
    float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float e = __builtin_fmaf(p, r2, r) + 1.0f;
    float s = __int_as_float(((int)n + 127) << 23);
    return e * s;
}

// vector_math.hpp:53-61 with the mul+add contraction nvcc applies written out explicitly
__device__ __forceinline__ float kt_dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.x, b.x, a.y * b.y)); }
__device__ __forceinline__ f3 kt_cross(f3 a, f3 b)
{
    f3 o;
    o.x = __builtin_fmaf(a.y, b.z, -(a.z * b.y));
    o.y = __builtin_fmaf(a.z, b.x, -(a.x * b.z));
    o.z = __builtin_fmaf(a.x, b.y, -(a.y * b.x));
    return o;
}
__device__ __forceinline__ f3 kt_sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 kt_add(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 kt_mul(const kt_mat33& m, f3 v)
{
    f3 o;
    o.x = __builtin_fmaf(m.m[2], v.z, __builtin_fmaf(m.m[0], v.x, m.m[1] * v.y));
    o.y = __builtin_fmaf(m.m[5], v.z, __builtin_fmaf(m.m[3], v.x, m.m[4] * v.y));
    o.z = __builtin_fmaf(m.m[8], v.z, __builtin_fmaf(m.m[6], v.x, m.m[7] * v.y));
    return o;
}
__device__ __forceinline__ f3 kt_normalized(f3 v)
{
    float inv = 1.0f / __builtin_sqrtf(kt_dot(v, v));  // rsqrtf restated
    return {v.x * inv, v.y * inv, v.z * inv};
}

// device.hpp:61-83
__device__ __forceinline__ short kt_pack_tsdf(float tsdf)
{
    int v = kt_f2i_rz(tsdf * KT_DIVISOR);
    v = v < -KT_DIVISOR ? -KT_DIVISOR : (v > KT_DIVISOR ? KT_DIVISOR : v);
    return (short)v;
}
// (float)v / 32767 (device.hpp:77-83), correctly rounded without the IEEE division sequence: q = v * RN(1/32767), one residual
// FMA and one correction FMA.  Bit-identical to the division for all 65536 shorts (tests/test_gpu_volume.py checks every one).
__device__ __forceinline__ float kt_unpack_tsdf(short v)
{
    const float x = (float)v, r = 1.0f / 32767.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 32767.0f, x), r, q);
}
#endif  // __HIPCC__
