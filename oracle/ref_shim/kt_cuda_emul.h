/*
 * kt_cuda_emul.h -- a minimal CUDA execution environment on the host CPU, so that the reference's OWN kernel
 * sources (/root/reference/src/frontend/cuda/{*.cu, containers/device_memory.cpp}) compile with a host C++ compiler and
 * run as a golden generator (oracle/_ref/libkt_ref.so).
 *
 * TEST INFRASTRUCTURE ONLY (same rule as oracle/kt_oracle.h): nothing under kintinuous_amd/, include/ or bench.py's
 * timed region may include, link or load this.  This file is the project's own code; no reference source is copied
 * into the repository -- the recipe (oracle/Makefile, target _ref) reads the sources where they lie.
 *
 * What is emulated
 *   - __global__ / __device__ / __shared__ / __constant__ qualifiers, dim3 + the vector types with CUDA's sizes and
 *     alignments, threadIdx / blockIdx / blockDim / gridDim, warpSize;
 *   - kernel<<<grid, block>>>(args): the recipe rewrites that ONE token sequence (not valid C++) on the fly to
 *     KT_REF_LAUNCH(grid, block, kernel(args)); every CUDA thread of a block runs as a fiber with its own stack, the
 *     blocks of a grid run one after the other;
 *   - __syncthreads() and the warp collectives (__shfl_down, __ballot, __all, __any, __syncwarp) as rendez-vous points
 *     between the fibers of a block / of a 32-lane warp -- divergence-free uses only (a collective reached by part of
 *     a warp is reported as a deadlock, not mis-executed);
 *   - atomicAdd / atomicInc (fibers are cooperatively scheduled on one OS thread), cudaMalloc* / cudaMemcpy* /
 *     cudaMemcpy{To,From}Symbol on host memory.
 *
 * Arithmetic: IEEE-754 binary32 with correctly rounded '/', sqrtf, denormals flushed inside kernels (--ftz=true, SSE
 * FTZ + DAZ); rsqrtf(x) = 1/sqrtf(x); __expf = libm expf;
 * __float2int_r{n,z,d} saturate and map NaN to 0 (CUDA's documented behaviour).  What the nvcc build of the reference
 * adds on top -- --prec-div=false --prec-sqrt=false and the approximate ex2/rsqrt units (CMakeLists.txt:47)
 * -- cannot be reproduced without the hardware; that is the remaining reference <-> _ref gap.
 * FMA contraction is left to the host compiler (-ffp-contract=fast), see oracle/Makefile.
 */
#ifndef KT_CUDA_EMUL_H_
#define KT_CUDA_EMUL_H_

#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include <float.h>
#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <type_traits>

#ifndef __CUDACC__
#define __CUDACC__ 1          /* containers/kernel_containers.hpp:43 keys GPU_HOST_DEVICE__ on it */
#endif
#ifndef __CUDA_ARCH__
#define __CUDA_ARCH__ 500     /* one of CMakeLists.txt:39's targets: native __shfl_down, __ldg, MAX_THREADS 1024 */
#endif

#define __global__
#define __device__
#define __host__
#define __shared__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

/* ---- vector types (CUDA vector_types.h sizes / alignments) ---- */
struct uchar3 { unsigned char x, y, z; };
struct alignas(4) uchar4 { unsigned char x, y, z, w; };
struct alignas(4) short2 { short x, y; };
struct short3 { short x, y, z; };
struct ushort3 { unsigned short x, y, z; };
struct alignas(8) int2 { int x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint3 { unsigned int x, y, z; };
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct double3 { double x, y, z; };

struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
    dim3(uint3 v) : x(v.x), y(v.y), z(v.z) {}
};

static inline uchar3 make_uchar3(unsigned char x, unsigned char y, unsigned char z) { uchar3 r = {x, y, z}; return r; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { uchar4 r = {x, y, z, w}; return r; }
static inline short2 make_short2(short x, short y) { short2 r = {x, y}; return r; }
static inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
static inline int3 make_int3(int x, int y, int z) { int3 r = {x, y, z}; return r; }
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }

/* ---- thread coordinates: written by the fiber scheduler on every switch ---- */
extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 32;

/* ---- runtime API subset ---- */
typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorMemoryAllocation = 2 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
enum cudaFuncCache { cudaFuncCachePreferNone = 0, cudaFuncCachePreferShared = 1, cudaFuncCachePreferL1 = 2 };

cudaError_t cudaMalloc(void** p, size_t n);
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t n) { return cudaMalloc((void**)p, n); }
cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t width_bytes, size_t height);
cudaError_t cudaFree(void* p);
cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind kind);
cudaError_t cudaMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t height, cudaMemcpyKind kind);
cudaError_t cudaMemset(void* p, int v, size_t n);
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "emulated CUDA runtime error"; }
template <class F> static inline cudaError_t cudaFuncSetCacheConfig(F, cudaFuncCache) { return cudaSuccess; }
#define cudaMemcpyToSymbol(sym, src, n) (memcpy((void*)&(sym), (const void*)(src), (n)), cudaSuccess)
#define cudaMemcpyFromSymbol(dst, sym, n) (memcpy((void*)(dst), (const void*)&(sym), (n)), cudaSuccess)

/* ---- launch: the recipe turns  k<<<g, b>>>(args);  into  KT_REF_LAUNCH(g, b, k(args));  ---- */
namespace ktemu {
void launch_impl(dim3 grid, dim3 block, void (*thread_body)(void*), void* closure);
template <class F> static inline void launch(dim3 grid, dim3 block, F&& f)
{
    typedef typename std::remove_reference<F>::type Fn;
    launch_impl(grid, block, [](void* p) { (*static_cast<Fn*>(p))(); }, (void*)&f);
}
void block_barrier();                                   /* __syncthreads */
unsigned warp_exchange(unsigned v, int src_lane);       /* every live lane of the warp deposits v, gets lane src_lane's */
unsigned warp_ballot(int pred);
unsigned lane_id();
}
#define KT_REF_LAUNCH(g, b, ...) ktemu::launch(dim3(g), dim3(b), [&]() { __VA_ARGS__; })

static inline void __syncthreads() { ktemu::block_barrier(); }
static inline void __syncwarp() { (void)ktemu::warp_ballot(0); }
static inline unsigned __ballot(int pred) { return ktemu::warp_ballot(pred); }
static inline int __all(int pred) { return ktemu::warp_ballot(!pred) == 0u; }
static inline int __any(int pred) { return ktemu::warp_ballot(pred) != 0u; }
/* shfl.down: a source lane at or beyond `width` returns the caller's own value */
static inline float __shfl_down(float v, int delta, int width = 32)
{
    unsigned u; memcpy(&u, &v, 4);
    int lane = (int)ktemu::lane_id();
    int src = ((lane % width) + delta < width) ? lane + delta : lane;
    u = ktemu::warp_exchange(u, src);
    memcpy(&v, &u, 4);
    return v;
}
static inline int __shfl_down(int v, int delta, int width = 32)
{
    int lane = (int)ktemu::lane_id();
    int src = ((lane % width) + delta < width) ? lane + delta : lane;
    return (int)ktemu::warp_exchange((unsigned)v, src);
}
template <typename T> static inline T __ldg(const T* p) { return *p; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicInc(unsigned* p, unsigned v) { unsigned o = *p; *p = (o >= v) ? 0u : o + 1u; return o; }

/* ---- device math with CUDA's documented semantics ---- */
static inline int ktemu_sat_i32(float r) { return (r != r) ? 0 : (r >= 2147483648.0f) ? 2147483647 : (r <= -2147483648.0f) ? (-2147483647 - 1) : (int)r; }
static inline int __float2int_rn(float x) { return ktemu_sat_i32(nearbyintf(x)); }   /* default rounding mode: ties to even */
static inline int __float2int_rz(float x) { return ktemu_sat_i32(truncf(x)); }
static inline int __float2int_rd(float x) { return ktemu_sat_i32(floorf(x)); }
static inline int __float2int_ru(float x) { return ktemu_sat_i32(ceilf(x)); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline float ktemu_expf(float x) { return expf(x); }
#define __expf ktemu_expf   /* glibc's <math.h> declares an internal symbol of that name */
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }

/* CUDA's global min / max overloads */
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, int b) { return fminf(a, (float)b); }
static inline float max(float a, int b) { return fmaxf(a, (float)b); }
static inline float min(int a, float b) { return fminf((float)a, b); }
static inline float max(int a, float b) { return fmaxf((float)a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
using std::isnan;
using std::abs;

#endif
