/*
 * Stand-in for the reference's frontend/cuda/warp.hpp when extract.cu is compiled for oracle/_ref (the recipe lists
 * this directory first on the quote-include path).  The original cannot be compiled by a host compiler: laneId() is
 * inline PTX (warp.hpp:75-80) and scan_warp (warp.hpp:41-63) is warp-synchronous code that is only correct when the 32
 * lanes execute every statement in lock step.  This restatement keeps the names and the results and spells the lock
 * step out as __syncwarp() rendez-vous points of the fiber emulation.  Only what extract.cu uses is provided.
 */
#ifndef UTILS_WARP_HPP_
#define UTILS_WARP_HPP_

enum ScanKind { exclusive, inclusive };

/* Hillis-Steele scan over the 32 entries of a warp (warp.hpp:41-63): each step reads ptr[idx - d] and ptr[idx] on all
 * lanes BEFORE any lane writes, which is what one SIMT instruction does. */
template <ScanKind Kind, class T>
__device__ __forceinline__ T scan_warp(volatile T* ptr, const unsigned int idx = threadIdx.x)
{
    const unsigned int lane = idx & 31;
    __syncwarp();
    for (unsigned int d = 1; d < 32; d <<= 1) {
        T s = ptr[idx];
        if (lane >= d) s = ptr[idx - d] + ptr[idx];
        __syncwarp();
        ptr[idx] = s;
        __syncwarp();
    }
    T r = (Kind == inclusive) ? ptr[idx] : ((lane > 0) ? ptr[idx - 1] : 0);
    __syncwarp();
    return r;
}

struct Warp
{
    enum { LOG_WARP_SIZE = 5, WARP_SIZE = 1 << LOG_WARP_SIZE, STRIDE = WARP_SIZE };
    static __device__ __forceinline__ unsigned int laneId() { return ktemu::lane_id(); }   /* warp.hpp:75-80: %laneid */
    static __device__ __forceinline__ unsigned int id()
    {
        int tid = threadIdx.z * blockDim.x * blockDim.y + threadIdx.y * blockDim.x + threadIdx.x;
        return tid >> LOG_WARP_SIZE;
    }
};

#endif
