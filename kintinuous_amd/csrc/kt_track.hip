// kt_track.hip -- tracking reductions for gfx950: projective point-to-plane ICP (a6), RGB-D photometric
// correspondence (a8) and its Jacobian reduction (a9), plus the device-side Gauss-Newton step (a7/a10:
// 6x6 LDL^T in double, Rodrigues, SE(3) update) that lets all iterations of a frame run back to back
// without a host round trip.  Reference: src/frontend/cuda/reduce.cu, src/frontend/ICPOdometry.cpp,
// RGBDOdometry.cpp, OdometryProvider.h.
//
// Reduction shape.  The 29 float sums are folded in EXACTLY the reference's order (reduce.cu:89-184, 321-339): 64 x 128
// "virtual threads" (vt) each accumulate pixels t, t + 8192, ... sequentially, a 32-lane __shfl_down tree folds each CUDA
// warp, the 4 warp sums of a block fold as (s0 + s2) + (s1 + s3), and the 64 block partials fold like
// reduceSum<<<1, 512>>> (two 32-lane trees, then s0 + s1).  A, b are therefore bit-identical to the oracle, which keeps
// whole trajectories and the fused volume bit-comparable.
// Only the 29 ADD chains per vt are inherently sequential; the per-pixel geometry (~220 VALU ops, two dependent gathers)
// is not.  So one 1024-thread workgroup per CUDA WARP (256 workgroups = every CU of the MI355X):
//   phase 1: its 32 vts' pixels are processed one per thread, fully parallel and coalesced (32 consecutive pixels per
//            k-step); the 7-vector row + inlier flag of every pixel goes to LDS as rows[k][component][vt] (bank-conflict free);
//   phase 2: thread (c, vt) owns product c of vt and walks k in order: acc = fma(row[a_c], row[b_c], acc) -- 29 x 32
//            independent chains, exactly the reference's per-thread sums (`sum.add(getProducts(i))`, reduce.cu:322-327: the
//            product's only user is the accumulation, so nvcc's -fmad=true contracts it; oracle/_ref, the reference source
//            compiled by clang with -ffp-contract=fast, pins that);
//   tree:    lanes are laid out c-major, so each 32-lane half-wave holds one product of the 32 vts = one CUDA warp:
//            ds_swizzle(xor 16) + DPP row_shl 8/4/2/1 reproduces warpReduceSum;
//   hand-off: the data is the flag (cdna_hip_programming.md G16 form R2): two warp sums go out as ONE aligned 8-byte granule, stored
//            write-through at agent scope; the last workgroup of the grid sweeps the 15 x 256 granules with agent-scope loads until
//            none of them is the sentinel, hands them back as sentinels, folds blocks and the final tree and, in the
//            device-resident path, solves the 6x6 system and updates the pose.  No ticket, no fence, no epoch.
#include "kt_internal.hpp"
#include "kt_setup.hpp"

#include <string.h>
#include <stdlib.h>
#include <float.h>
#include <math.h>

// ------------------------------------------------------------------------------------------------
// block / grid reduction
// ------------------------------------------------------------------------------------------------
#define KT_RED_THREADS 1024  // one workgroup per CUDA warp of the reference's <<<64, 128>>> launch (ICPOdometry.cpp:123-124)
#define KT_RED_BLOCKS 256    // 64 CUDA blocks x 4 warps
#define KT_VT_TOTAL 8192     // 64 x 128 virtual threads
#define KT_KBATCH 40         // k-steps staged in LDS per pass: 40 x 8 x 32 floats = 40 KB
#define KT_RED_SLOTS 32      // 29 used

// warpReduceSum over one CUDA-warp-sized group (32 lanes), reduce.cu:89-129.  Lane 0 of each group ends with
// ((..(v0 + v16) + (v8 + v24)) ..) exactly as the reference's tree; the other lanes are don't-care.
// offset 16: ds_swizzle (lane ^ 16 inside each 32-lane half); offsets 8,4,2,1: DPP row_shl inside the 16-lane row.
template <int CTRL>
__device__ __forceinline__ float kt_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float kt_warp32_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // and 0x1f, or 0, xor 0x10
    v += kt_dpp<0x108>(v);  // row_shl:8  (lane l reads lane l + 8)
    v += kt_dpp<0x104>(v);
    v += kt_dpp<0x102>(v);
    v += kt_dpp<0x101>(v);
    return v;
}

// product c (0..27) of the 7-vector row = row[KT_PA[c]] * row[KT_PB[c]]  (JtJJtrSE3 field order, internal.h:98-149)
__device__ __constant__ unsigned char KT_PA[28] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6};
__device__ __constant__ unsigned char KT_PB[28] = {0, 1, 2, 3, 4, 5, 6, 1, 2, 3, 4, 5, 6, 2, 3, 4, 5, 6, 3, 4, 5, 6, 4, 5, 6, 5, 6, 6};

// The reduction in two parts.  kt_reduce29_publish: phases 1 and 2 and the warp tree of every workgroup, ending with its 29
// granules stored; RowFn(i, row[7]) -> found computes one pixel.  kt_reduce29_sweep: the last workgroup gathers the grid sums into
// total[0..28] (LDS).  kt_reduce29 = both, returning true (workgroup-uniformly) in the sweeping workgroup.
struct kt_no_prefetch { __device__ __forceinline__ void operator()() const {} };
// Hand-off granules (round 4, third form).  ONE aligned 8-byte word carries TWO of a workgroup's 29 warp sums -- products 2p and 2p + 1,
// which the two half-waves of wave p hold -- and "the data is the flag": a granule is valid when it is not the SENTINEL (all ones: two
// NaNs of a payload no arithmetic produces; a publisher that did hold 0xffffffff stores 0xfffffffe).  The sweeping wave writes the
// sentinel back into every granule it has consumed; the next launch that publishes into the buffer is behind this one on the stream,
// so the kernel boundary orders the two.  A launch that runs several iterations (kt_icp_level_kernel) uses two sets in turn and has no
// boundary between the hand-back of iteration `it` and the publish of iteration it + 2 into the same words (advisor, round 5: the pair was
// ordered only by a causal chain of relaxed accesses).  There the chain is made explicit: every sweeping wave waits for its sentinel
// stores to COMPLETE (they are write-through agent-scope stores: completed = performed at the memory side; s_waitcnt vmcnt(0), paid by
// the fifteen sweeping waves -- 1..15 -- which have nothing else to do while wave 0 solves) and arrives at a workgroup barrier; the solving
// wave, which sweeps nothing and so has no such stores of its own, passes the same barrier immediately before it stores the pose granules.  So: hand-back performed -> barrier -> pose stored -> pose observed by a
// publisher's polling wave -> its workgroup's barrier -> that workgroup's stores of iterations it + 1, it + 2.  Cost against the unordered
// form, same box: 3775 against 3790 frames/s.  (Measured alternatives, profiles/r06_experiments.md: one set per iteration -- +0.26 us per
// iteration, the sets go cold in the memory-side cache; hand-back by the publishers' own lanes -- +0.5 us, 3840 scattered 8-byte stores per
// iteration instead of 60 coalesced 512-byte ones; wave 0 sweeping as well and waiting for its own stores in the middle of its tail -- +0.1 us.)
// (Rounds 2-3 tagged every 4-byte sum with a 4-byte epoch: twice the lines for the sweep to
// fetch, and the sweep -- agent-scope loads go past the L2, one compute unit issues all of them -- is paid per LINE REQUEST:
// profiles/r04_experiments.md.)  Place of workgroup wg's (= CUDA block wg / 4, warp wg % 4) granule of pair p: the four warps of a block
// are 64 granules apart, so that the sweeping wave's q-th load -- lane b takes warp q of block b -- reads 512 contiguous bytes.
#define KT_RED_PAIRS 15
#define KT_GRANULE_SENTINEL 0xffffffffffffffffull
// kt_icp_level_kernel: its two granule sets, behind everything else in the context's hand-off buffer (u64 indices; [0, 8192) and
// [8192, 16384): the two kt_reduce29 sets, 16384..: the residual launch's words, kt_residual_granules / kt_residual_partials)
#define KT_LEVEL_SETS 4                // (kt_icp_level_kernel uses the first two; kt_joint_level_kernel two per reduction)
#define KT_LEVEL_SET_STRIDE 4096      // >= KT_RED_PAIRS * KT_RED_BLOCKS = 3840
#define KT_LEVEL_SET_BASE 32768
#define KT_POSE_ABORT 15              // pose_gran[15]: {0, seq} of the iteration whose sweep gave up (same 128-byte line as the 12 pose granules)
__device__ __forceinline__ int kt_granule_index(int pair, int wg) { return pair * KT_RED_BLOCKS + (wg & 3) * (KT_RED_BLOCKS / 4) + (wg >> 2); }
// lane 0 of every wave p < 15 publishes {sum of product 2p (lanes 0..31), sum of product 2p + 1 (lanes 32..63)}
__device__ __forceinline__ void kt_publish_pair(unsigned long long* __restrict__ granules, float wsum)
{
    unsigned int lo = __float_as_uint(wsum), hi = (unsigned int)__builtin_amdgcn_readlane((int)__float_as_uint(wsum), 32);
    lo = lo == 0xffffffffu ? 0xfffffffeu : lo;
    const int pair = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0 && pair < KT_RED_PAIRS)
        __hip_atomic_store(&granules[kt_granule_index(pair, blockIdx.x)], ((unsigned long long)hi << 32) | (unsigned long long)lo, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
typedef float kt_rows_t[8][32];   // LDS staging rows[k][component][vt], KT_KBATCH of them

template <typename RowFn>
__device__ __forceinline__ void kt_reduce29_publish(const RowFn& fn, int n, unsigned long long* __restrict__ granules, kt_rows_t* rows)
{
    KT_TS(0);
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 32;           // first virtual thread of this CUDA warp
    const int comp = tid >> 5, vt = tid & 31;  // phase-2 role
    const int my_a = comp < 28 ? KT_PA[comp] : 7, my_b = comp < 28 ? KT_PB[comp] : 7;
    // number of pixels of virtual thread t0 + vt: i = t, t + 8192, ... < n
    const int nk_vt = (n - (t0 + vt) + KT_VT_TOTAL - 1) / KT_VT_TOTAL;
    const int nk_blk = (n - t0 + KT_VT_TOTAL - 1) / KT_VT_TOTAL;
    float acc = 0.f;
    for (int kb = 0; kb < nk_blk; kb += KT_KBATCH) {
        const int kcount = min(KT_KBATCH, nk_blk - kb);
        // phase 1: one pixel per thread and pass, 32 consecutive pixels per k.  Two passes are fused (pixels p and p + 1024 of the
        // batch): RowFn is branch-free, so the loads of both pixels are in flight together.
        for (int p = tid; p < kcount * 32; p += 2 * KT_RED_THREADS) {
            const int p2 = p + KT_RED_THREADS;
            const int kl = p >> 5, v = p & 31, kl2 = p2 >> 5;
            const int i = t0 + v + (kb + kl) * KT_VT_TOTAL;
            const int i2 = t0 + v + (kb + kl2) * KT_VT_TOTAL;       // (p2 & 31) == v
            const bool has2 = p2 < kcount * 32;
            float row[7], row2[7];
            const bool found = fn(min(i, n - 1), row) && i < n;
            if (__builtin_amdgcn_ballot_w64(has2) != 0) {            // wave-uniform: most waves have no second pixel
                const bool found2 = fn(min(i2, n - 1), row2) && i2 < n && has2;
                if (has2) {
#pragma unroll
                    for (int q = 0; q < 7; ++q) rows[kl2][q][v] = found2 ? row2[q] : 0.0f;
                    rows[kl2][7][v] = found2 ? 1.0f : 0.0f;
                }
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) rows[kl][q][v] = found ? row[q] : 0.0f;
            rows[kl][7][v] = found ? 1.0f : 0.0f;
        }
        __syncthreads();
        // phase 2: thread (comp, vt) accumulates its product in k order (the reference's per-thread sum.add)
        if (comp < 29) {
            const int kend = min(kcount, nk_vt - kb);
            if (comp < 28) {
                for (int kl = 0; kl < kend; ++kl) acc = __builtin_fmaf(rows[kl][my_a][vt], rows[kl][my_b][vt], acc);
            } else {
                for (int kl = 0; kl < kend; ++kl) acc += rows[kl][7][vt];
            }
        }
        __syncthreads();
    }
    KT_TS(1);
    // warp tree: the 32 lanes of a half-wave hold product `comp` of the 32 virtual threads of this CUDA warp
    const float wsum = kt_warp32_sum(acc);   // (threads of a product past the 29th hold 0)
    kt_publish_pair(granules, wsum);
    KT_TS(2);
}

// Two reductions over the same pixel range in ONE pass (the joint RGB-D + ICP iteration): phase 1 computes both rows of a pixel with
// the loads of both in flight together, phase 2 walks both products of (comp, vt).  Every sum is formed exactly as by two
// kt_reduce29_publish calls; only the waiting is shared.
template <typename RowFnA, typename RowFnB>
__device__ __forceinline__ void kt_reduce29_publish2(const RowFnA& fa, const RowFnB& fb, int n, unsigned long long* __restrict__ granules_a,
                                                     unsigned long long* __restrict__ granules_b, kt_rows_t* rows_a, kt_rows_t* rows_b)
{
    KT_TS(0);
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 32;
    const int comp = tid >> 5, vt = tid & 31;
    const int my_a = comp < 28 ? KT_PA[comp] : 7, my_b = comp < 28 ? KT_PB[comp] : 7;
    const int nk_vt = (n - (t0 + vt) + KT_VT_TOTAL - 1) / KT_VT_TOTAL;
    const int nk_blk = (n - t0 + KT_VT_TOTAL - 1) / KT_VT_TOTAL;
    float acc_a = 0.f, acc_b = 0.f;
    for (int kb = 0; kb < nk_blk; kb += KT_KBATCH) {
        const int kcount = min(KT_KBATCH, nk_blk - kb);
        for (int p = tid; p < kcount * 32; p += KT_RED_THREADS) {
            const int kl = p >> 5, v = p & 31;
            const int i = t0 + v + (kb + kl) * KT_VT_TOTAL;
            // both rows in stages, the loads of a stage issued together: two exposed memory latencies per pixel instead of four
            const int ii = min(i, n - 1);
            const auto term = fb.fetch_term(ii);
            f3 vcurr, ncurr, vcurr_g, vprev_g, nprev_g;
            bool inimg;
            int g;
            fa.fetch_curr(ii, vcurr, ncurr);
            typename RowFnB::taps tp;
            fb.fetch_taps(term, tp);
            fa.project(vcurr, vcurr_g, inimg, g);
            fa.fetch_prev(g, vprev_g, nprev_g);
            {
                float rb[7];   // the short one first: its row is parked in LDS before the long one's arithmetic starts
                const bool found_b = fb.finish(term, tp, rb) && i < n;
#pragma unroll
                for (int q = 0; q < 7; ++q) rows_b[kl][q][v] = found_b ? rb[q] : 0.0f;
                rows_b[kl][7][v] = found_b ? 1.0f : 0.0f;
            }
            float ra[7];
            const bool found_a = fa.finish(ncurr, vcurr_g, vprev_g, nprev_g, inimg, ra) && i < n;
#pragma unroll
            for (int q = 0; q < 7; ++q) rows_a[kl][q][v] = found_a ? ra[q] : 0.0f;
            rows_a[kl][7][v] = found_a ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (comp < 29) {
            const int kend = min(kcount, nk_vt - kb);
            if (comp < 28) {
                for (int kl = 0; kl < kend; ++kl) {
                    acc_a = __builtin_fmaf(rows_a[kl][my_a][vt], rows_a[kl][my_b][vt], acc_a);
                    acc_b = __builtin_fmaf(rows_b[kl][my_a][vt], rows_b[kl][my_b][vt], acc_b);
                }
            } else {
                for (int kl = 0; kl < kend; ++kl) { acc_a += rows_a[kl][7][vt]; acc_b += rows_b[kl][7][vt]; }
            }
        }
        __syncthreads();
    }
    KT_TS(1);
    const float wsum_a = kt_warp32_sum(acc_a), wsum_b = kt_warp32_sum(acc_b);
    kt_publish_pair(granules_a, wsum_a);
    kt_publish_pair(granules_b, wsum_b);
    KT_TS(2);
}

// NS reductions published by the same launch are swept together: every wave issues the loads of all its granules -- 4 warp pairs x NS
// sets -- before it looks at any of them, so the sweep costs one memory round trip, not one per set.
// bound of the sweep's spin (a test hook lowers it: kt_debug_handoff_fault); read once per sweep, next to the first granule loads
__device__ unsigned int kt_sweep_spin_limit = 1u << 22;
// ... and the bound that decides in practice: ticks of the constant 100 MHz clock (s_memrealtime) a wait may last.  A look costs an agent-scope
// round trip (~1.5 us), so 2^22 looks would be seconds; 50 ms is three orders of magnitude above the longest legitimate wait (a workgroup kept
// out of its compute unit by the tracker's own side streams: < 1 ms) and short enough for the tracker to fall back to the stepwise chain
// within the frame (kt_tracker.hip: complete_frame).  kt_debug_wait_limit changes it.
__device__ unsigned int kt_wait_limit_ticks = 5000000u;
__device__ __forceinline__ unsigned long long kt_ticks() { return __builtin_amdgcn_s_memrealtime(); }

// patience: multiplier of kt_wait_limit_ticks.  A launch whose workgroups wait for each other INSIDE it (kt_icp_level_kernel) must give up soon
// -- while it waits it holds the compute units somebody else's workgroups may need to make the progress it is waiting for; the stream-ordered
// kernels hold ONE compute unit while their sweep waits and nothing circular can involve them, so they wait 40 times longer (2 s): long enough
// for another process's level launch to run into ITS bound and get out of the way (tests/test_gpu_two_process.py).
template <int NS>
__device__ __forceinline__ void kt_reduce29_sweep_n(unsigned long long* const (&granules)[NS], float* const (&total)[NS], unsigned int patience = 40u)
{
    const int tid = threadIdx.x;
    KT_TS(3);
    __shared__ unsigned int timed_out;
    if (tid == 0) timed_out = 0;
    __syncthreads();
    // The sweeping workgroup.  blockReduceSum second stage (reduce.cu:131-164) per CUDA block b: lanes 0..3 hold the warp sums,
    // lanes 4..31 zero; offsets 16, 8, 4 add zeros, offset 2 gives s0+s2 and s1+s3, offset 1 adds them.  Then
    // reduceSum<<<1, 512>>> (reduce.cu:166-184): threads 0..63 hold 0 + in[b]; warps 0 and 1 fold with the 32-lane tree, the rest
    // is zero; the final first-warp tree reduces to s0 + s1.  Wave w < 15 handles the pair of products (2w, 2w + 1); lane = CUDA block b,
    // whose 4 warp pairs are the granules kt_granule_index(w, 4b + q), q = 0..3: 64 granules apart, so that load q of a wave is one
    // contiguous run.  Each wave re-reads its granules until none of them is the sentinel (bounded: a hand-off that never completes
    // raises slot 31 of total[], which the callers report as an error), then hands them back as sentinels for the next launch.
    {
        // (waves 1..15 sweep: wave 0 -- the one that solves -- issues no hand-back stores, so it has none to wait for before it publishes a
        // pose in kt_icp_level_kernel)
        const int w = (tid >> 6) - 1, lane = tid & 63;
        if (w >= 0 && w < KT_RED_PAIRS) {   // (wave-uniform)
            unsigned long long g[NS][4];
            bool ok;
            unsigned int spins = 0;
            const unsigned int spin_limit = *(volatile const unsigned int*)&kt_sweep_spin_limit;
            const unsigned long long tick_limit = (unsigned long long)*(volatile const unsigned int*)&kt_wait_limit_ticks * patience;
            unsigned long long tick0 = 0;   // the clock is read from the 16th look on: a hand-off that completes at once never pays for it
            for (;;) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const unsigned long long* pp = &granules[s][w * KT_RED_BLOCKS + lane];   // kt_granule_index(w, 4 lane + q) = pp + 64 q
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[s][q] = __hip_atomic_load(pp + q * (KT_RED_BLOCKS / 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ok = true;
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok && g[s][q] != KT_GRANULE_SENTINEL;
                if (__all(ok) || ++spins > spin_limit) break;
                if ((spins & 15u) == 0) {
                    const unsigned long long now = kt_ticks();
                    if (spins == 16u) tick0 = now;
                    else if (now - tick0 > tick_limit) break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const bool all_ok = __all(ok);
            if (all_ok) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __hip_atomic_store(&granules[s][w * KT_RED_BLOCKS + lane + q * (KT_RED_BLOCKS / 4)], KT_GRANULE_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int c = 2 * w + r;
                    const float s0 = __uint_as_float((unsigned int)(g[s][0] >> (32 * r))) + 0.0f;
                    const float s1 = __uint_as_float((unsigned int)(g[s][1] >> (32 * r))) + 0.0f;
                    const float s2 = __uint_as_float((unsigned int)(g[s][2] >> (32 * r))) + 0.0f;
                    const float s3 = __uint_as_float((unsigned int)(g[s][3] >> (32 * r))) + 0.0f;
                    const float blk = (s0 + s2) + (s1 + s3);
                    const float tr = kt_warp32_sum(0.0f + blk);
                    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tr), 0));
                    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tr), 32));
                    if (lane == 0 && c < 29) total[s][c] = (lo + 0.0f) + (hi + 0.0f);
                }
            if (lane == 0 && !all_ok) atomicOr(&timed_out, 1u);
        }
    }
    __syncthreads();
    if (tid == 0) total[NS - 1][KT_RED_SLOTS - 1] = timed_out ? 1.0f : 0.0f;  // slot 31: the hand-off never completed (reported by the callers)
    __syncthreads();
    KT_TS(4);
}

__device__ __forceinline__ void kt_reduce29_sweep(unsigned long long* __restrict__ granules, float (&total)[KT_RED_SLOTS], unsigned int patience = 40u)
{
    unsigned long long* const gs[1] = {granules};
    float* const ts[1] = {total};
    kt_reduce29_sweep_n<1>(gs, ts, patience);
}

// `pre` runs in the sweeping workgroup between its publish and the sweep: the place to issue loads the epilogue will need
// (their latency hides under the sweep).
// (A dedicated extra workgroup that only sweeps -- polling while the others are still in their pixel loops -- was measured in round 3:
// odometry stage 204.9 us against 204.0 with the last publishing workgroup sweeping, frame rate -1.5 %: the sweep is not waiting for
// its own pixel loop but for the slowest of the other 255.)
#define KT_RED_GRID KT_RED_BLOCKS
__device__ __forceinline__ bool kt_red_publishes() { return true; }
__device__ __forceinline__ bool kt_red_sweeps() { return blockIdx.x == KT_RED_GRID - 1; }

template <typename RowFn, typename PreFn = kt_no_prefetch>
__device__ __forceinline__ bool kt_reduce29(const RowFn& fn, int n, unsigned long long* __restrict__ granules, float (&total)[KT_RED_SLOTS],
                                            const PreFn& pre = PreFn())
{
    __shared__ kt_rows_t rows[KT_KBATCH];
    if (kt_red_publishes()) kt_reduce29_publish(fn, n, granules, rows);
    if (!kt_red_sweeps()) return false;
    pre();
    kt_reduce29_sweep(granules, total);
    return true;
}

// Every granule back to the sentinel, on the context's stream (ordered behind everything enqueued so far).  A sweep that gave up leaves
// the buffer in an undefined state: it hands nothing back, and the publishers it did not wait for store AFTER it has looked -- values the
// NEXT launch's sweep could take for its own (advisor, round 4; with the epoch tags of rounds 2-3 a stale granule could not be mistaken).
// So EVERY path that reports a time-out to its caller refills before the next reduction launch: kt_icp_step / kt_rgb_step /
// kt_rgb_residual / kt_icp_track here, complete_frame and kt_tracker_reset in kt_tracker.hip (tests/test_gpu_track.py forces one).
int kt_refill_granules(kt_ctx* c)
{
    KT_HIP(hipMemsetAsync(c->red_partials, 0xff, sizeof(double) * 32 * c->red_max_blocks, c->stream));
    return KT_OK;
}

// epoch of the next launch on this context (only the host form of the residual launch still tags its granules with it): never 0, never
// all ones -- the buffer's fill pattern, which a never-written residual granule would otherwise share with launch 2^32 - 1 --, never
// repeated between two refills of the buffer
static unsigned int kt_next_epoch(kt_ctx* c)
{
    if (++c->red_epoch >= 0xffffffffu) {
        (void)kt_refill_granules(c);
        c->red_epoch = 1;
    }
    return c->red_epoch;
}
// the residual kernel's granules live behind the two [32][256] kt_reduce29 blocks in the same buffer
static unsigned long long* kt_residual_granules(kt_ctx* c) { return (unsigned long long*)c->red_partials + 64 * KT_RED_BLOCKS; }
// second kt_reduce29 granule block, for kernels that run two reductions (kt_joint_kernel)
// ... and behind those the plain per-workgroup words of the device-path residual launch ([2][256] unsigned int)
static unsigned int* kt_residual_partials(kt_ctx* c) { return (unsigned int*)((unsigned long long*)c->red_partials + 64 * KT_RED_BLOCKS + 1024); }
static unsigned long long* kt_second_granules(kt_ctx* c) { return (unsigned long long*)c->red_partials + 32 * KT_RED_BLOCKS; }

// ------------------------------------------------------------------------------------------------
// a6  icpStep -> icpKernel + reduceSum                reduce.cu:186-419
// ------------------------------------------------------------------------------------------------
struct kt_icp_args {
    const float* vmap_curr; const float* nmap_curr;
    const float* vmap_g_prev; const float* nmap_g_prev;
    kt_intr intr;
    int cols, rows;
    float dist_thres, angle_thres;
    // the two thresholds on the SQUARES (kt_icp_set_thresholds): sqrtf is correctly rounded and monotone, so "sqrtf(x) <= T" is "x <= X" for
    // the largest float X whose root is <= T, and "sqrtf(x) < A" is "x <= X'" for the largest float whose root is < A -- found once per
    // launch on the host, instead of two 16-instruction IEEE square roots per pixel (reduce.cu:247-253)
    float dist2_le, sine2_le;
    // pose: either immediate (host path) or read from the device state (device-resident path)
    kt_mat33 Rcurr; float tcurr[3];
    kt_mat33 Rprev_inv; float tprev[3];
    kt_track_state* state;     // nullptr on the host path
    int first;                 // device path, first iteration of a frame: the pose comes from the fields above (== previous pose) and
                               // the epilogue initialises *state (no host-to-device copy of the state per frame)
    unsigned long long* granules; unsigned int fault;   // inter-workgroup hand-off (kt_reduce29); fault: test hook, below
    float* out29;              // host path: 29 floats
    int mode;                  // KT_MODE_*
    int keep29;                // KT_MODE_ICP_SOLVE: also leave the 29 sums in state->icp29 (kt_icp_track's last iteration: the caller's A)
    // kt_icp_level_kernel (round 5): n_iter Gauss-Newton iterations of ONE pyramid level in ONE launch.  The pose goes from an iteration's
    // solving workgroup to the others as 12 granules {float, seq} tagged seq0 + iteration; Rcurr / tcurr then carry the frame's PREVIOUS pose
    // (also the starting pose of the frame's first launch), Rprev_inv / tprev as always.  level_gran: two granule sets, iteration `it` of the
    // launch reduces through set it & 1 (see the hand-off comment at the top).  pose_gran[KT_POSE_ABORT]: {., seq} of the iteration whose
    // sweep gave up.
    int n_iter; unsigned int seq0; unsigned long long* pose_gran; unsigned long long* level_gran;
    // round 6: ALL pyramid levels of a frame in one launch (two kernel boundaries and their first loads less per frame).  lv[0 .. n_levels) in
    // the order they run (coarse to fine); n_iter above = the sum of theirs; the single-level fields above (maps, intr, cols, rows) are
    // those of lv[0] and are re-pointed by the kernel at every level switch.
    int n_levels;
    struct level { const float* vmap_curr; const float* nmap_curr; const float* vmap_g_prev; const float* nmap_g_prev; kt_intr intr; int cols, rows, n_iter; } lv[KT_LEVELS];
};

struct kt_icp_row {
    // what a pixel's row reads of the launch: the maps and the image geometry of the pyramid level (own copies, wave-uniform: kt_icp_level_kernel
    // re-points them at every level switch), the two thresholds on the squares
    struct view { const float* vmap_curr; const float* nmap_curr; const float* vmap_g_prev; const float* nmap_g_prev; kt_intr intr; int cols, rows; float dist2_le, sine2_le; } a;
    kt_mat33 Rcurr, Rprev_inv;
    f3 tcurr, tprev;
    __device__ __forceinline__ explicit kt_icp_row(const kt_icp_args& k)
        : a{k.vmap_curr, k.nmap_curr, k.vmap_g_prev, k.nmap_g_prev, k.intr, k.cols, k.rows, k.dist2_le, k.sine2_le} {}
    __device__ __forceinline__ void set_level(const kt_icp_args::level& v)
    {
        a.vmap_curr = v.vmap_curr; a.nmap_curr = v.nmap_curr; a.vmap_g_prev = v.vmap_g_prev; a.nmap_g_prev = v.nmap_g_prev;
        a.intr = v.intr; a.cols = v.cols; a.rows = v.rows;
    }
    // kt_icp_level_kernel: the current frame's vertex / normal of pixel pf_i, requested before the pose of the previous iteration arrived
    int pf_i = -1;
    f3 pf_v = {0.f, 0.f, 0.f}, pf_n = {0.f, 0.f, 0.f};
    // search() + getProducts(), reduce.cu:213-277, for pixel i; fills row[7] (zeros when no correspondence), returns found.
    // Branch-free: the reference's early returns become predicates and the gather index of a rejected pixel is clamped to 0, so
    // two calls inlined back to back have all their loads issued together (one exposed latency instead of two).
    // In four stages, so that a caller with a second row to compute (kt_reduce29_publish2) can put the loads of both behind each other:
    // fetch_curr -> project -> fetch_prev -> finish.  operator() is the four in sequence.
    __device__ __forceinline__ void fetch_curr(int i, f3& vcurr, f3& ncurr) const
    {
        const int plane = a.cols * a.rows;
        vcurr = {a.vmap_curr[i], a.vmap_curr[i + plane], a.vmap_curr[i + 2 * plane]};  // y * cols + x == i
        ncurr = {a.nmap_curr[i], a.nmap_curr[i + plane], a.nmap_curr[i + 2 * plane]};
    }
    __device__ __forceinline__ void project(const f3& vcurr, f3& vcurr_g, bool& inimg, int& g) const
    {
        const int cols = a.cols, rows = a.rows;
        vcurr_g = kt_add(kt_mul(Rcurr, vcurr), tcurr);
        const f3 vcurr_cp = kt_mul(Rprev_inv, kt_sub(vcurr_g, tprev));
        const int ux = kt_f2i_rn(vcurr_cp.x * a.intr.fx / vcurr_cp.z + a.intr.cx);
        const int uy = kt_f2i_rn(vcurr_cp.y * a.intr.fy / vcurr_cp.z + a.intr.cy);
        inimg = !(ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp.z < 0);
        g = inimg ? (int)kt_mad24((unsigned int)uy, (unsigned int)cols, (unsigned int)ux) : 0;   // (uy, ux < 2^24 inside the image)
    }
    __device__ __forceinline__ void fetch_prev(int g, f3& vprev_g, f3& nprev_g) const
    {
        const int plane = a.cols * a.rows;
        vprev_g = {a.vmap_g_prev[g], a.vmap_g_prev[g + plane], a.vmap_g_prev[g + 2 * plane]};
        nprev_g = {a.nmap_g_prev[g], a.nmap_g_prev[g + plane], a.nmap_g_prev[g + 2 * plane]};
    }
    __device__ __forceinline__ bool finish(const f3& ncurr, const f3& vcurr_g, const f3& vprev_g, const f3& nprev_g, bool inimg, float (&row)[7]) const
    {
        const f3 ncurr_g = kt_mul(Rcurr, ncurr);
        const f3 dv = kt_sub(vprev_g, vcurr_g);
        const float dist2 = kt_dot(dv, dv);                       // dist = sqrtf(dist2) <= dist_thres   <=>  dist2 <= dist2_le
        const f3 cr = kt_cross(ncurr_g, nprev_g);
        const float sine2 = kt_dot(cr, cr);                       // sine = sqrtf(sine2) < angle_thres   <=>  sine2 <= sine2_le
        const bool found = inimg && (sine2 <= a.sine2_le && dist2 <= a.dist2_le && !kt_isnan(ncurr.x) && !kt_isnan(nprev_g.x));
        const f3 s_cp = kt_mul(Rprev_inv, kt_sub(vcurr_g, tprev));
        const f3 d_cp = kt_mul(Rprev_inv, kt_sub(vprev_g, tprev));
        const f3 n_cp = kt_mul(Rprev_inv, nprev_g);
        const f3 sxn = kt_cross(s_cp, n_cp);
        row[0] = found ? n_cp.x : 0.0f; row[1] = found ? n_cp.y : 0.0f; row[2] = found ? n_cp.z : 0.0f;
        row[3] = found ? sxn.x : 0.0f; row[4] = found ? sxn.y : 0.0f; row[5] = found ? sxn.z : 0.0f;
        row[6] = found ? kt_dot(n_cp, kt_sub(s_cp, d_cp)) : 0.0f;
        return found;
    }
    __device__ __forceinline__ bool operator()(int i, float (&row)[7]) const
    {
        f3 vcurr, ncurr, vcurr_g, vprev_g, nprev_g;
        bool inimg;
        int g;
        if (i == pf_i) { vcurr = pf_v; ncurr = pf_n; }
        else fetch_curr(i, vcurr, ncurr);
        project(vcurr, vcurr_g, inimg, g);
        fetch_prev(g, vprev_g, nprev_g);
        return finish(ncurr, vcurr_g, vprev_g, nprev_g, inimg, row);
    }
};

// X = the largest float with sqrtf(X) <= T (strict = 0) or sqrtf(X) < T (strict = 1); -1 when no x >= 0 qualifies (then "x <= X" is false for
// every sum of squares, and for NaN, as the comparison of the root was).  sqrtf on the host is the correctly rounded IEEE root, like the
// device's __builtin_sqrtf it replaces; a NaN threshold makes every comparison false on both sides.  tests/test_host_logic.py walks the
// neighbours of X for a spread of thresholds.
extern "C" float kt_debug_sq_threshold(float T, int strict)
{
    if (!(T >= 0.0f) || (strict && !(T > 0.0f))) return -1.0f;
    if (T == INFINITY) return strict ? FLT_MAX : INFINITY;
    const auto ok = [&](float x) { const float r = sqrtf(x); return strict ? r < T : r <= T; };
    float x = T * T;
    if (x == INFINITY) x = FLT_MAX;
    while (x > 0.0f && !ok(x)) x = nextafterf(x, -INFINITY);
    if (!ok(x)) return -1.0f;   // (strict, T the smallest denormal: not even 0 ... sqrtf(0) = 0 < T holds, so this cannot happen; kept for clarity)
    while (x < FLT_MAX && ok(nextafterf(x, INFINITY))) x = nextafterf(x, INFINITY);
    return x;
}
static void kt_icp_set_thresholds(kt_icp_args& a, float dist_thres, float angle_thres)
{
    a.dist_thres = dist_thres; a.angle_thres = angle_thres;
    a.dist2_le = kt_debug_sq_threshold(dist_thres, 0);
    a.sine2_le = kt_debug_sq_threshold(angle_thres, 1);
}

__global__ __launch_bounds__(KT_RED_THREADS) void kt_icp_kernel(const kt_icp_args a)
{
    if (a.fault && blockIdx.x == 0) return;   // test hook (kt_debug_handoff_fault): a publisher that never arrives -> the sweep must give up
    kt_icp_row fn(a);
    if (a.state && !a.first) {
        // pose produced by the previous iteration's epilogue (kernel boundary orders the accesses)
        for (int k = 0; k < 9; ++k) { fn.Rcurr.m[k] = a.state->Rcurr[k]; fn.Rprev_inv.m[k] = a.state->Rprev_inv[k]; }
        fn.tcurr = {a.state->tcurr[0], a.state->tcurr[1], a.state->tcurr[2]};
        fn.tprev = {a.state->tprev[0], a.state->tprev[1], a.state->tprev[2]};
    } else {
        fn.Rcurr = a.Rcurr; fn.Rprev_inv = a.Rprev_inv;
        fn.tcurr = {a.tcurr[0], a.tcurr[1], a.tcurr[2]};
        fn.tprev = {a.tprev[0], a.tprev[1], a.tprev[2]};
    }
    __shared__ float total[KT_RED_SLOTS];
    kt_pose_stage ps;
    const bool solve_here = a.mode == KT_MODE_ICP_SOLVE;
    auto pre = [&]() { if (solve_here && !a.first) ps.fetch(a.state); };
    if (!kt_reduce29(fn, a.cols * a.rows, a.granules, total, pre)) return;
    __shared__ double sys[KT_SYS_DOUBLES], pose_d[KT_POSE_STAGE_DOUBLES], tail_work[KT_TAIL_WORK_DOUBLES];
    __shared__ float pose_f[KT_POSE_STAGE_FLOATS];
    if (solve_here) {   // ICPOdometry.cpp:127-128: the float sums widened into the double system, one element per lane
        if (threadIdx.x < 42) sys[threadIdx.x] = (double)total[kt_sys_slot(threadIdx.x)];
        if (!a.first) ps.park(pose_d, pose_f);
        else if (threadIdx.x == 64) {   // ICPOdometry.cpp:70-85: identity increment, the previous pose from the kernel arguments
#pragma unroll
            for (int k = 0; k < 16; ++k) pose_d[k] = (k % 5 == 0) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) pose_f[k] = a.Rcurr.m[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) pose_f[9 + k] = a.tprev[k];
        }
        __syncthreads();
    }
    if (a.mode == KT_MODE_HOST) {
        if (threadIdx.x < KT_RED_SLOTS && (threadIdx.x < 29 || threadIdx.x == KT_RED_SLOTS - 1)) a.out29[threadIdx.x] = total[threadIdx.x];
    } else if (a.mode == KT_MODE_ICP_SOLVE) {
        // ICPOdometry.cpp:127-178: the solve and the pose update on the lanes of the first wave (kt_solve_and_update_wave); the
        // bookkeeping stores on a lane of the second
        if (threadIdx.x == 64) {
            a.state->last_residual[0] = total[27];
            a.state->last_residual[1] = total[28];
            if (a.keep29)
                for (int k = 0; k < 29; ++k) a.state->icp29[k] = total[k];
            if (a.first) {  // ICPOdometry.cpp:70-85: previous pose and its inverse
#pragma unroll
                for (int k = 0; k < 9; ++k) { a.state->Rprev[k] = a.Rcurr.m[k]; a.state->Rprev_inv[k] = a.Rprev_inv.m[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) a.state->tprev[k] = a.tprev[k];
                a.state->handoff_timeout = total[KT_RED_SLOTS - 1] != 0.0f ? 1 : 0;
            } else if (total[KT_RED_SLOTS - 1] != 0.0f) {
                a.state->handoff_timeout = 1;
            }
        }
        if (threadIdx.x < 64) {
            kt_solve_and_update_wave(a.state, sys, pose_d, pose_f, tail_work);
#ifdef KT_ICP_TIMING
            if (threadIdx.x == 0) { const unsigned long long t5 = wall_clock64(); for (int q = 0; q < 7; ++q) a.state->icp29[q] = (float)(kt_ts[q + 0] - kt_ts[0]); a.state->icp29[7] = (float)(t5 - kt_ts[0]); }
#endif
        }
    } else if (threadIdx.x == 0) {  // KT_MODE_ICP_STASH: joint RGB-D + ICP, the rgb kernel's epilogue combines and solves
        for (int k = 0; k < 29; ++k) a.state->icp29[k] = total[k];
        if (total[KT_RED_SLOTS - 1] != 0.0f) a.state->handoff_timeout = 1;
    }
}


// ------------------------------------------------------------------------------------------------
// Several iterations of a level in one launch (round 5).  Of a stream-ordered iteration's 9 us, 1.45 are the kernel boundary and ~0.7 the
// first loads of the next launch; here the 256 workgroups stay resident for all iterations of the level: after publishing its reduction
// granules a workgroup requests the current-frame values of its first pixel for the NEXT iteration (they do not depend on the pose) and
// polls the 12 pose granules the solving workgroup publishes at the end of its tail -- measured hand-over latency 0.6-0.7 us
// (profiles/r05_icp_overlap/).  The solving workgroup (the last one, as in kt_icp_kernel) keeps the accumulated increment in LDS from one
// iteration to the next and takes its own pose from LDS.  Consecutive iterations use the two granule sets in turn.  The arithmetic of an
// iteration is kt_icp_kernel's, operation for operation (same row functor, same reduction, same tail).
// Unlike two kernels on two streams (the overlapped chain that was measured and dropped), one kernel cannot starve itself: all of its
// workgroups are dispatched before any of them waits for more than the first pose.
// ------------------------------------------------------------------------------------------------
// Wave 0 of a publishing workgroup waits for the pose of the iteration tagged `want`: 12 granules tagged with its sequence number, and -- lane 12 --
// the abort granule.  Bounded (looks, and `patience` x twice the sweep's time bound: a sweep that succeeds at the edge of ITS bound must still find
// its pollers waiting).  s_pose[0 .. 12) receive the pose, s_pose[12] != 0: it never came, or the launch was aborted at or after that iteration.
__device__ __forceinline__ void kt_level_wait_pose(const kt_icp_args& a, unsigned int want, float* s_pose, unsigned int patience = 1u)
{
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned int spin_limit = patience > 1u ? 0xffffffffu : min(*(volatile const unsigned int*)&kt_sweep_spin_limit, 1u << 20);
    const unsigned long long tick_limit = 2ull * *(volatile const unsigned int*)&kt_wait_limit_ticks * patience;
    unsigned long long tick0 = 0;   // (read from the 16th look on, as in the sweep)
    unsigned long long g = 0;
    unsigned int spins = 0;
    bool ok, gone;
    for (;;) {
        if (lane < 13) g = __hip_atomic_load(&a.pose_gran[lane == 12 ? KT_POSE_ABORT : lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ahead = (int)((unsigned int)(g >> 32) - want);
        ok = lane >= 12 || ahead == 0;
        gone = lane == 12 && ahead >= 0 && ahead < a.n_iter;   // the launch was aborted at or after the iteration waited for
        if (__all(ok) || __any(gone) || ++spins > spin_limit) break;
        if ((spins & 15u) == 0) {
            const unsigned long long now = kt_ticks();
            if (spins == 16u) tick0 = now;
            else if (now - tick0 > tick_limit) break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    const bool got = __all(ok) && !__any(gone);
    if (lane < 12) s_pose[lane] = __uint_as_float((unsigned int)g);
    if (lane == 12) s_pose[12] = got ? 0.0f : 1.0f;
}

// The set-up's arguments are the launch's SECOND kernel argument, read from the kernel-argument segment where they are used: as ordinary by-value
// uses their ~70 scalars were loaded at the kernel's entry and lived -- spilled -- across the whole iteration loop (15 VGPRs to scratch in the hot
// path).  The address is laundered so that no load can be hoisted above this point.
__device__ __forceinline__ const kt_setup_args* kt_level_setup_args()
{
    static_assert(alignof(kt_setup_args) == 8 && alignof(kt_icp_args) == 8, "kernel-argument layout: the second argument starts at the first one's size rounded up to 8");
    // (the OFFSET is laundered, not the pointer: the address stays in the constant address space, so the fields arrive through the scalar cache like
    // any kernel argument -- as flat loads of a laundered generic pointer they were vector reads of the host-visible argument buffer, microseconds each)
    unsigned int off = (unsigned int)((sizeof(kt_icp_args) + 7ull) & ~7ull);
    asm volatile("" : "+s"(off));
    typedef __attribute__((address_space(4))) const char* kt_karg_ptr;
    kt_karg_ptr k = (kt_karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    return (const kt_setup_args*)(k + off);
}

#ifdef KT_ICP_LEVEL_WAVES   // A/B builds: waves per SIMD the register budget is cut for (8: 64 VGPRs = two workgroups per compute unit)
__global__ __launch_bounds__(KT_RED_THREADS, KT_ICP_LEVEL_WAVES) void kt_icp_level_kernel(const kt_icp_args a, const kt_setup_args su)
#else
__global__ __launch_bounds__(KT_RED_THREADS) void kt_icp_level_kernel(const kt_icp_args a, const kt_setup_args su)
#endif
{
    if (a.fault && blockIdx.x == 0) return;   // test hook (kt_debug_handoff_fault)
    __shared__ float total[KT_RED_SLOTS];
    __shared__ double sys[KT_SYS_DOUBLES], pose_d[KT_POSE_STAGE_DOUBLES], tail_work[KT_TAIL_WORK_DOUBLES];
    __shared__ float pose_f[KT_POSE_STAGE_FLOATS];
    __shared__ float s_pose[13];   // the pose of the iteration about to run (+ [12] != 0: it never came)
    const bool sweeper = kt_red_sweeps();
    kt_icp_row fn(a);
    fn.Rprev_inv = a.Rprev_inv;
    fn.tprev = {a.tprev[0], a.tprev[1], a.tprev[2]};
    if (a.first) {   // ICPOdometry.cpp:70-85: the frame starts from the previous pose
        fn.Rcurr = a.Rcurr;
        fn.tcurr = {a.tprev[0], a.tprev[1], a.tprev[2]};
    } else {         // the pose the previous launch left (the kernel boundary orders the accesses)
        if (a.state->handoff_timeout) return;   // an earlier launch of this frame gave up: the frame has no pose, nothing here is worth waiting for
        for (int k = 0; k < 9; ++k) fn.Rcurr.m[k] = a.state->Rcurr[k];
        fn.tcurr = {a.state->tcurr[0], a.state->tcurr[1], a.state->tcurr[2]};
    }
    if (sweeper) {   // the solving workgroup's carry: the accumulated increment and the frame's previous pose
        if (threadIdx.x < 16) pose_d[threadIdx.x] = a.first ? ((threadIdx.x % 5 == 0) ? 1.0 : 0.0) : a.state->resultRt[threadIdx.x];
        else if (threadIdx.x < 25) pose_f[threadIdx.x - 16] = a.Rcurr.m[threadIdx.x - 16];
        else if (threadIdx.x < 28) pose_f[threadIdx.x - 16] = a.tprev[threadIdx.x - 25];
    }
    if (threadIdx.x == 12) s_pose[12] = 0.0f;
    int it = 0;   // iteration of the launch, counted across the levels (sequence numbers, granule set parity)
    bool aborted = false;
    for (int L = 0; L < a.n_levels; ++L) {
    fn.set_level(a.lv[L]);
    const int n = fn.a.cols * fn.a.rows;
    // the first pixel this thread is asked for in every iteration (kt_reduce29_publish: p = tid of batch 0)
    const int t0 = blockIdx.x * 32, nk_blk = (n - t0 + KT_VT_TOTAL - 1) / KT_VT_TOTAL;
    const int pf_i = ((int)threadIdx.x < min(KT_KBATCH, nk_blk) * 32) ? min(t0 + ((int)threadIdx.x & 31) + ((int)threadIdx.x >> 5) * KT_VT_TOTAL, n - 1) : -1;
    for (int lit = 0; lit < a.lv[L].n_iter; ++lit, ++it) {
        unsigned long long* const gran = a.level_gran + (size_t)(it & 1) * KT_LEVEL_SET_STRIDE;
        if (it > 0) {
            fn.pf_i = -1;
            if (!sweeper) {
                if (pf_i >= 0) { fn.fetch_curr(pf_i, fn.pf_v, fn.pf_n); fn.pf_i = pf_i; }   // in flight while the pose is awaited
                // Wait for the pose of iteration it - 1 (kt_level_wait_pose).  A workgroup whose wait gives up, or that finds the launch aborted (the
                // sweeping workgroup's own wait gave up: granule KT_POSE_ABORT carries the iteration), LEAVES the kernel: it publishes nothing
                // further, so the sweep of the next iteration cannot complete and the time-out is reported by the one workgroup that writes the
                // state -- an error cannot be lost between two writers (advisor, round 5: a poller's `handoff_timeout = 1` could be overwritten
                // by the sweeper's first-iteration store).
                if (threadIdx.x < 64) kt_level_wait_pose(a, a.seq0 + (unsigned int)it - 1u, s_pose);
            }
            __syncthreads();   // (the solving workgroup: its tail wrote s_pose)
            if (s_pose[12] != 0.0f) return;   // workgroup-uniform: no pose (see above)
            // (wave-uniform values: into scalar registers, where kt_icp_kernel's arguments live too -- as per-lane copies they cost 12 VGPRs)
            const auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
            for (int k = 0; k < 9; ++k) fn.Rcurr.m[k] = uni(s_pose[k]);
            fn.tcurr = {uni(s_pose[9]), uni(s_pose[10]), uni(s_pose[11])};
        }
        __shared__ kt_rows_t rows[KT_KBATCH];
        kt_reduce29_publish(fn, n, gran, rows);
        if (!sweeper) continue;
        kt_reduce29_sweep(gran, total, 1u);
        // ICPOdometry.cpp:127-178, as kt_icp_kernel's KT_MODE_ICP_SOLVE epilogue
        if (threadIdx.x < 42) sys[threadIdx.x] = (double)total[kt_sys_slot(threadIdx.x)];
        const bool timed_out = total[KT_RED_SLOTS - 1] != 0.0f;   // (workgroup-uniform: read behind the sweep's closing barrier)
        __syncthreads();
        if (threadIdx.x == 64) {
            a.state->last_residual[0] = total[27];
            a.state->last_residual[1] = total[28];
            if (a.first && it == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) { a.state->Rprev[k] = a.Rcurr.m[k]; a.state->Rprev_inv[k] = a.Rprev_inv.m[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) a.state->tprev[k] = a.tprev[k];
                a.state->handoff_timeout = timed_out ? 1 : 0;
            } else if (timed_out) {
                a.state->handoff_timeout = 1;
            }
            // a sweep that gave up ends the launch: the waiting workgroups are told which iteration it was and leave, and so does this one
            // (the frame has no pose: the set-up kernel parks its fusion, the host re-runs its odometry in the stepwise form)
            if (timed_out)
                __hip_atomic_store(&a.pose_gran[KT_POSE_ABORT], (unsigned long long)(a.seq0 + (unsigned int)it) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (timed_out) { aborted = true; goto kt_level_done; }   // (workgroup-uniform)
        // hand-back performed before the pose leaves (see the hand-off comment at the top of the file): waves 1..15 wait for their stores and
        // arrive at the barrier now; wave 0 passes it inside its tail, right in front of the pose granules' stores
        if (threadIdx.x < 64) kt_solve_and_update_wave(a.state, sys, pose_d, pose_f, tail_work, a.pose_gran, a.seq0 + (unsigned int)it, s_pose);
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    }
    // ---- the frame's set-up in the launch's epilogue (round 6; csrc/kt_setup.hpp).  Until now the 255 publishing workgroups left after their last
    // granules and a launch of its own (kt_frame_setup_kernel: a kernel boundary, 300-odd small workgroups, 8 us in the frame) turned the result
    // into what the fusion kernels read.  Here every workgroup waits for the FINAL pose like for any other (the solving workgroup needs nothing
    // from anybody at this point: the wait cannot be part of a cycle), takes its share of the carry and checkpoint blocks, and the solving
    // workgroup's first two waves are block 0 -- pose, shift decision, plan check, z tables, the host's mirror.
kt_level_done:
    const kt_setup_args* sp = kt_level_setup_args();
    if (aborted) {
        // (only the sweeping workgroup gets here.  Fused set-up: the host waits for the mirror -- it is told that the frame has no pose; the
        // fusion kernels are parked)
        if (sp->fused && threadIdx.x < 128) {
            float Rl[9], tl[3];
            for (int k = 0; k < 9; ++k) Rl[k] = a.Rcurr.m[k];
            for (int k = 0; k < 3; ++k) tl[k] = a.tprev[k];
            kt_setup_block0(*sp, Rl, tl, 1, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
        }
        return;
    }
    if (sp->fused) {
        if (!sweeper) {
            if (threadIdx.x < 64) kt_level_wait_pose(a, a.seq0 + (unsigned int)a.n_iter - 1u, s_pose, 40u);
        }
        __syncthreads();
        if (s_pose[12] != 0.0f) return;
        float R[9], tv[3];
        for (int k = 0; k < 9; ++k) R[k] = s_pose[k];
        for (int k = 0; k < 3; ++k) tv[k] = s_pose[9 + k];
        const int nside = sp->carry_groups + sp->walk_groups;
        for (int vb = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 8); vb < nside; vb += KT_RED_GRID * 4)
            kt_setup_side_block(*sp, vb + 1, (int)(threadIdx.x & 255), R, tv);
        if (sweeper && threadIdx.x < 128) kt_setup_block0(*sp, R, tv, 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
    }
}

int kt_icp_launch(kt_ctx* c, kt_icp_args& a, const kt_setup_args* fused_setup = nullptr);
int kt_icp_launch(kt_ctx* c, kt_icp_args& a, const kt_setup_args* fused_setup)
{
    a.granules = (unsigned long long*)c->red_partials;
    a.fault = 0;
    if (c->fault_skip > 0) --c->fault_skip;
    else if (c->fault_count > 0) { --c->fault_count; a.fault = 1; }
    if (a.n_iter > 0) {
        static_assert(KT_LEVEL_SET_STRIDE >= KT_RED_PAIRS * KT_RED_BLOCKS, "a granule set per iteration");
        if ((size_t)KT_LEVEL_SET_BASE + (size_t)KT_LEVEL_SETS * KT_LEVEL_SET_STRIDE > (size_t)32 * c->red_max_blocks) {
            kt_set_error("kt_icp_level_kernel: the hand-off buffer is too small");
            return KT_ERR_ARG;
        }
        a.level_gran = (unsigned long long*)c->red_partials + KT_LEVEL_SET_BASE;
        a.pose_gran = c->pose_gran;
        // (Through hipLaunchCooperativeKernel -- the runtime's own promise of co-residency -- the default run measured 1858 frames/s against 3720:
        // a cooperative launch is serialised against everything else on the device, profiles/r06_experiments.md.  Residency is checked once per
        // device instead (kt_icp_levels_fit), every wait inside the launch is bounded in time, and a launch that gives up costs the frame a
        // re-run, not an error: kt_tracker.hip complete_frame.)
        kt_setup_args su;
        if (fused_setup) su = *fused_setup;
        else { memset(&su, 0, sizeof(su)); su.fused = 0; }
        hipLaunchKernelGGL(kt_icp_level_kernel, dim3(KT_RED_GRID), dim3(KT_RED_THREADS), 0, c->stream, a, su);
    } else {
        hipLaunchKernelGGL(kt_icp_kernel, dim3(KT_RED_GRID), dim3(KT_RED_THREADS), 0, c->stream, a);
    }
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// host unpack of the 29 sums, reduce.cu:401-418
static void kt_unpack29_host(const float* h, float* A, float* b, float* residual)
{
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = h[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    if (residual) { residual[0] = h[27]; residual[1] = h[28]; }
}

extern "C" int kt_icp_step(kt_ctx* c, const kt_mat33* Rcurr, const float tcurr[3], const float* vmap_curr, const float* nmap_curr,
                           const kt_mat33* Rprev_inv, const float tprev[3], const kt_intr* intr, const float* vmap_g_prev,
                           const float* nmap_g_prev, int cols, int rows, float dist_thres, float angle_thres, float* A_host,
                           float* b_host, float* residual_host)
{
    KT_ARG(c && Rcurr && tcurr && vmap_curr && nmap_curr && Rprev_inv && tprev && intr && vmap_g_prev && nmap_g_prev && A_host && b_host);
    KT_ARG(cols > 0 && rows > 0);
    kt_icp_args a;
    a.vmap_curr = vmap_curr; a.nmap_curr = nmap_curr; a.vmap_g_prev = vmap_g_prev; a.nmap_g_prev = nmap_g_prev;
    a.intr = *intr; a.cols = cols; a.rows = rows; kt_icp_set_thresholds(a, dist_thres, angle_thres);
    a.Rcurr = *Rcurr; a.Rprev_inv = *Rprev_inv;
    for (int k = 0; k < 3; ++k) { a.tcurr[k] = tcurr[k]; a.tprev[k] = tprev[k]; }
    a.state = nullptr; a.first = 0; a.out29 = c->red_out; a.mode = KT_MODE_HOST; a.keep29 = 0;
    a.n_iter = 0; a.seq0 = 0; a.pose_gran = nullptr; a.level_gran = nullptr; a.n_levels = 0;
    int s = kt_icp_launch(c, a);
    if (s != KT_OK) return s;
    KT_HIP(hipMemcpyAsync(c->red_out_host, c->red_out, sizeof(float) * KT_RED_SLOTS, hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    if (c->red_out_host[KT_RED_SLOTS - 1] != 0.0f) { (void)kt_refill_granules(c); kt_set_error("icpStep: inter-workgroup hand-off timed out"); return KT_ERR_STATE; }
    kt_unpack29_host(c->red_out_host, A_host, b_host, residual_host);
    return KT_OK;
}

int kt_icp_step_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr,
                       const float* vmap_g_prev, const float* nmap_g_prev, int cols, int rows, float dist_thres, float angle_thres,
                       int mode, const kt_track_state* init, int keep29)
{
    kt_icp_args a;
    a.n_iter = 0; a.seq0 = 0; a.pose_gran = nullptr; a.level_gran = nullptr; a.n_levels = 0;
    a.vmap_curr = vmap_curr; a.nmap_curr = nmap_curr; a.vmap_g_prev = vmap_g_prev; a.nmap_g_prev = nmap_g_prev;
    a.intr = *intr; a.cols = cols; a.rows = rows; kt_icp_set_thresholds(a, dist_thres, angle_thres);
    a.state = state; a.out29 = nullptr; a.mode = mode;
    a.first = 0;
    a.keep29 = keep29;
    if (init) {  // first iteration of a frame (ICP-only path): the starting pose travels in the kernel arguments
        if (mode != KT_MODE_ICP_SOLVE) { kt_set_error("kt_icp_step_device: init needs KT_MODE_ICP_SOLVE"); return KT_ERR_ARG; }
        a.first = 1;
        memcpy(a.Rcurr.m, init->Rcurr, sizeof(a.Rcurr.m));
        memcpy(a.Rprev_inv.m, init->Rprev_inv, sizeof(a.Rprev_inv.m));
        for (int k = 0; k < 3; ++k) { a.tcurr[k] = init->tcurr[k]; a.tprev[k] = init->tprev[k]; }
    }
    return kt_icp_launch(c, a);
}

// The Gauss-Newton iterations of n_levels pyramid levels (in the order given: coarse to fine) in ONE launch (kt_icp_level_kernel).  frame: Rprev,
// tprev, Rprev_inv of the frame (the init record of the stepwise form); first: the frame's first launch (the starting pose is the previous
// pose, the state is initialised).
int kt_icp_levels_device(kt_ctx* c, kt_track_state* state, int n_levels, const float* const* vmaps_curr, const float* const* nmaps_curr, const kt_intr* intrs,
                         const float* const* vmaps_g_prev, const float* const* nmaps_g_prev, const int* cols, const int* rows, const int* n_iter,
                         float dist_thres, float angle_thres, const kt_track_state* frame, int first, const kt_setup_args* fused_setup)
{
    KT_ARG(n_levels >= 1 && n_levels <= KT_LEVELS);
    kt_icp_args a;
    memset(&a, 0, sizeof(a));
    int total = 0, used = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (n_iter[l] <= 0) continue;
        kt_icp_args::level& v = a.lv[used++];
        v.vmap_curr = vmaps_curr[l]; v.nmap_curr = nmaps_curr[l]; v.vmap_g_prev = vmaps_g_prev[l]; v.nmap_g_prev = nmaps_g_prev[l];
        v.intr = intrs[l]; v.cols = cols[l]; v.rows = rows[l]; v.n_iter = n_iter[l];
        total += n_iter[l];
    }
    if (total <= 0) return KT_OK;
    a.n_levels = used;
    a.vmap_curr = a.lv[0].vmap_curr; a.nmap_curr = a.lv[0].nmap_curr; a.vmap_g_prev = a.lv[0].vmap_g_prev; a.nmap_g_prev = a.lv[0].nmap_g_prev;
    a.intr = a.lv[0].intr; a.cols = a.lv[0].cols; a.rows = a.lv[0].rows; kt_icp_set_thresholds(a, dist_thres, angle_thres);
    a.state = state; a.out29 = nullptr; a.mode = KT_MODE_ICP_SOLVE; a.first = first ? 1 : 0; a.keep29 = 0;
    memcpy(a.Rcurr.m, frame->Rprev, sizeof(a.Rcurr.m));
    memcpy(a.Rprev_inv.m, frame->Rprev_inv, sizeof(a.Rprev_inv.m));
    for (int k = 0; k < 3; ++k) { a.tcurr[k] = frame->tprev[k]; a.tprev[k] = frame->tprev[k]; }
    a.n_iter = total;
    if (c->odo_seq > 0xfffff000u) c->odo_seq = 0;   // (2^32 iterations are 16 hours at 3800 frames/s)
    a.seq0 = c->odo_seq + 1u;
    c->odo_seq += (unsigned int)total;
    return kt_icp_launch(c, a, fused_setup);
}
// ... one level (the form round 5 launched three times per frame)
int kt_icp_level_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr, const float* vmap_g_prev,
                        const float* nmap_g_prev, int cols, int rows, float dist_thres, float angle_thres, const kt_track_state* frame, int first, int n_iter)
{
    return kt_icp_levels_device(c, state, 1, &vmap_curr, &nmap_curr, intr, &vmap_g_prev, &nmap_g_prev, &cols, &rows, &n_iter, dist_thres, angle_thres, frame, first, nullptr);
}

// ICPOdometry::getIncrementalTransformation (ICPOdometry.cpp:68-186) as ONE entry point (SURVEY 8(b) export list): pose in / pose out,
// every Gauss-Newton iteration of every pyramid level enqueued back to back with the 6x6 solve, Rodrigues and the SE(3) update in
// the reduction kernel's epilogue (kt_solve_and_update) -- no host round trip per iteration (the reference: 19 x {2 launches,
// cudaDeviceSynchronize, 116-byte copy, Eigen LDLT on the host}).  The same kernels and the same state machine as the tracker's
// odometry stage; the tracking state lives in the context.
extern "C" int kt_icp_track(kt_ctx* c, const float* const vmaps_curr[KT_LEVELS], const float* const nmaps_curr[KT_LEVELS],
                            const float* const vmaps_g_prev[KT_LEVELS], const float* const nmaps_g_prev[KT_LEVELS], int cols, int rows,
                            const kt_intr* intr, const kt_mat33* Rprev, const float tprev[3], const int iterations[KT_LEVELS], float dist_thres,
                            float angle_thres, kt_mat33* Rcurr_out, float tcurr_out[3], float A_last_out[36], float residual_out[2])
{
    KT_ARG(c && vmaps_curr && nmaps_curr && vmaps_g_prev && nmaps_g_prev && intr && Rprev && tprev && iterations && Rcurr_out && tcurr_out && cols > 0 && rows > 0);
    if (!c->track_state) {
        KT_HIP(hipMalloc((void**)&c->track_state, sizeof(kt_track_state)));
        KT_HIP(hipMemsetAsync(c->track_state, 0, sizeof(kt_track_state), c->stream));
    }
    kt_track_state* st = (kt_track_state*)c->track_state;
    kt_track_state init;
    memset(&init, 0, sizeof(init));
    memcpy(init.Rprev, Rprev->m, sizeof(init.Rprev)); memcpy(init.Rcurr, Rprev->m, sizeof(init.Rcurr));
    memcpy(init.tprev, tprev, sizeof(init.tprev)); memcpy(init.tcurr, tprev, sizeof(init.tcurr));
    kt_mat33_inverse(init.Rprev, init.Rprev_inv);  // ICPOdometry.cpp:81
    int total = 0, done = 0;
    for (int l = 0; l < KT_LEVELS; ++l) { KT_ARG(iterations[l] >= 0 && (iterations[l] == 0 || (vmaps_curr[l] && nmaps_curr[l] && vmaps_g_prev[l] && nmaps_g_prev[l]))); total += iterations[l]; }
    for (int l = KT_LEVELS - 1; l >= 0; --l) {
        const int div = 1 << l;
        const kt_intr li = {intr->fx / div, intr->fy / div, intr->cx / div, intr->cy / div};   // Intr::operator(), internal.h:255-259
        for (int it = 0; it < iterations[l]; ++it, ++done)
            KT_TRY(kt_icp_step_device(c, st, vmaps_curr[l], nmaps_curr[l], &li, vmaps_g_prev[l], nmaps_g_prev[l], cols >> l, rows >> l, dist_thres, angle_thres,
                                      KT_MODE_ICP_SOLVE, done == 0 ? &init : nullptr, done == total - 1));
    }
    if (total == 0) {   // no iterations: the pose stays the previous one
        *Rcurr_out = *Rprev;
        for (int k = 0; k < 3; ++k) tcurr_out[k] = tprev[k];
        if (A_last_out) memset(A_last_out, 0, 36 * sizeof(float));
        if (residual_out) residual_out[0] = residual_out[1] = 0.0f;
        return KT_OK;
    }
    kt_track_state out;
    KT_HIP(hipMemcpyAsync(&out, st, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    if (out.handoff_timeout) { (void)kt_refill_granules(c); kt_set_error("kt_icp_track: inter-workgroup hand-off timed out"); return KT_ERR_STATE; }
    memcpy(Rcurr_out->m, out.Rcurr, sizeof(out.Rcurr));
    memcpy(tcurr_out, out.tcurr, sizeof(out.tcurr));
    if (A_last_out) {
        float b[6], r[2];
        kt_unpack29_host(out.icp29, A_last_out, b, r);
    }
    if (residual_out) { residual_out[0] = out.last_residual[0]; residual_out[1] = out.last_residual[1]; }
    return KT_OK;
}

// ------------------------------------------------------------------------------------------------
// a8  computeRgbResidual -> residualKernel            reduce.cu:668-864
// int2 {count, sum diff^2} reduction: integer sums are order-independent, one atomic pair per wave.
// ------------------------------------------------------------------------------------------------
struct kt_residual_args {
    float min_scale;
    const int16_t* dIdx; const int16_t* dIdy;
    const float* last_depth; const float* next_depth;
    const uint8_t* last_image; const uint8_t* next_image;
    kt_dataterm* corres;
    float max_depth_delta;
    float kt_[3]; kt_mat33 krkinv;
    kt_track_state* state;   // device path: krkinv / kt read from the state, sigma written back
    int cols, rows;
    int* out2;               // host path: {count, sigma} written by the sweeping workgroup
    unsigned long long* granules; unsigned int epoch;   // host path: [2][gridDim.x] {epoch, value} hand-off granules (as in kt_reduce29)
    unsigned int* partials;  // device path: [2][KT_RES_MAX_BLOCKS] plain per-workgroup sums, read by the next launch (kt_residual_sigma)
    const uint8_t* cand;     // tracker path: the pose-independent half of the per-pixel test, precomputed per frame (kt_residual_candidates_kernel)
    int write_all;           // with cand: 0 = only candidates are stored (the other DataTerms were zeroed by an earlier iteration of this frame)
};

// The part of residualKernel's per-pixel test that does not depend on the pose (reduce.cu:686-716): image border, the 4x4 window of
// non-zero intensities, the gradient threshold and a valid depth.
__device__ __forceinline__ bool kt_residual_candidate(const uint8_t* __restrict__ next_image, const int16_t* __restrict__ dIdx,
                                                      const int16_t* __restrict__ dIdy, const float* __restrict__ next_depth, int cols, int rows,
                                                      float min_scale, int i, int j0)
{
    if (!(j0 < cols - 5 && i < rows - 1)) return false;
    bool valid = true;
    for (int u = max(i - 2, 0); u < min(i + 2, rows); u++)
        for (int v = max(j0 - 2, 0); v < min(j0 + 2, cols); v++) valid = valid && (next_image[u * cols + v] > 0);
    if (!valid) return false;
    const int valx = dIdx[i * cols + j0], valy = dIdy[i * cols + j0];
    const float mTwo = (float)((valx * valx) + (valy * valy));
    if (!(mTwo >= min_scale)) return false;
    return !kt_isnan(next_depth[i * cols + j0]);
}

__global__ __launch_bounds__(256) void kt_residual_candidates_kernel(const uint8_t* __restrict__ next_image, const int16_t* __restrict__ dIdx,
                                                                     const int16_t* __restrict__ dIdy, const float* __restrict__ next_depth,
                                                                     int cols, int rows, float min_scale, uint8_t* __restrict__ cand)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cols * rows) return;
    const int i = k / cols, j0 = k - i * cols;
    cand[k] = kt_residual_candidate(next_image, dIdx, dIdy, next_depth, cols, rows, min_scale, i, j0) ? 1 : 0;
}

int kt_rgb_residual_candidates(kt_ctx* c, float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* next_depth,
                               const uint8_t* next_image, int cols, int rows, uint8_t* cand)
{
    hipLaunchKernelGGL(kt_residual_candidates_kernel, dim3(kt_div_up(cols * rows, 256)), dim3(256), 0, c->stream, next_image, dIdx, dIdy,
                       next_depth, cols, rows, min_scale, cand);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// residualKernel's per-pixel body (reduce.cu:718-760) for pixel k of the NEXT image, given the pose-independent half of its test (`candidate`):
// the DataTerm, and whether it counts (valid); the loads that do not depend on the pose (depth and intensity of this pixel) are the caller's
__device__ __forceinline__ bool kt_residual_pixel(const kt_residual_args& a, const float (&K)[9], const float (&kt)[3], int k, int cols, int rows, bool candidate,
                                                  float d1, uint8_t ni, kt_dataterm& corres)
{
    corres.zero_x = corres.zero_y = corres.one_x = corres.one_y = 0;
    corres.diff = 0.f;
    corres.valid = 0;
    corres.pad[0] = corres.pad[1] = corres.pad[2] = 0;
    const int y = k / cols, x = k - y * cols;
    const float xf = (float)x, yf = (float)y;
    const float transformed_d1 = __builtin_fmaf(d1, __builtin_fmaf(K[6], xf, K[7] * yf) + K[8], kt[2]);
    const int u0 = kt_f2i_rn(__builtin_fmaf(d1, __builtin_fmaf(K[0], xf, K[1] * yf) + K[2], kt[0]) / transformed_d1);
    const int v0 = kt_f2i_rn(__builtin_fmaf(d1, __builtin_fmaf(K[3], xf, K[4] * yf) + K[5], kt[1]) / transformed_d1);
    const bool inimg = candidate && u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows;
    const int gi = inimg ? v0 * cols + u0 : 0;
    const float d0 = a.last_depth[gi];
    const uint8_t li = a.last_image[gi];
    if (inimg && d0 > 0 && fabsf(transformed_d1 - d0) <= a.max_depth_delta && li != 0) {
        corres.zero_x = (int16_t)u0; corres.zero_y = (int16_t)v0;
        corres.one_x = (int16_t)x; corres.one_y = (int16_t)y;
        corres.diff = (float)ni - (float)li;
        corres.valid = 1;
        return true;
    }
    return false;
}

#define KT_RES_THREADS 1024
#define KT_RES_MAX_BLOCKS 256   // <= 4 granules per sweeping lane and sum
static int kt_residual_blocks(int cols, int rows) { const int g = kt_div_up(cols * rows, KT_RES_THREADS); return g > KT_RES_MAX_BLOCKS ? KT_RES_MAX_BLOCKS : g; }
template <bool PRE>
__global__ __launch_bounds__(KT_RES_THREADS) void kt_residual_kernel(const kt_residual_args a)
{
    const int cols = a.cols, rows = a.rows, n = cols * rows;
    float K[9], kt[3];
    if (a.state) {
        for (int q = 0; q < 9; ++q) K[q] = a.state->krkinv[q];
        for (int q = 0; q < 3; ++q) kt[q] = a.state->kt[q];
    } else {
        for (int q = 0; q < 9; ++q) K[q] = a.krkinv.m[q];
        for (int q = 0; q < 3; ++q) kt[q] = a.kt_[q];
    }
    int cnt = 0;
    unsigned int sig = 0;  // wraps modulo 2^32 like the reference's int sum
    for (int k = blockIdx.x * KT_RES_THREADS + threadIdx.x; k < n; k += gridDim.x * KT_RES_THREADS) {
        const int i = k / cols, j0 = k - i * cols;
        kt_dataterm corres;
        corres.zero_x = corres.zero_y = corres.one_x = corres.one_y = 0;
        corres.diff = 0.f;
        corres.valid = 0;
        corres.pad[0] = corres.pad[1] = corres.pad[2] = 0;
        // The reference's nested tests (reduce.cu:718-760) as predicates: the loads that do not depend on the pose (candidate flag, depth
        // and intensity of this pixel) go out together, the two gathers at the projected pixel together behind them -- two exposed memory
        // latencies per pixel instead of four; the index of a pixel that fails a test is clamped to 0.
        const float d1 = a.next_depth[k];
        const uint8_t ni = a.next_image[k];
        const bool candidate = PRE ? a.cand[k] != 0
                                   : kt_residual_candidate(a.next_image, a.dIdx, a.dIdy, a.next_depth, cols, rows, a.min_scale, i, j0);
        if (__builtin_amdgcn_ballot_w64(candidate) != 0) {   // wave-uniform: no candidate among these 64 pixels, no arithmetic
            const int y = i, x = j0;
            const float xf = (float)x, yf = (float)y;
            const float transformed_d1 = __builtin_fmaf(d1, __builtin_fmaf(K[6], xf, K[7] * yf) + K[8], kt[2]);
            const int u0 = kt_f2i_rn(__builtin_fmaf(d1, __builtin_fmaf(K[0], xf, K[1] * yf) + K[2], kt[0]) / transformed_d1);
            const int v0 = kt_f2i_rn(__builtin_fmaf(d1, __builtin_fmaf(K[3], xf, K[4] * yf) + K[5], kt[1]) / transformed_d1);
            const bool inimg = candidate && u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows;
            const int gi = inimg ? v0 * cols + u0 : 0;
            const float d0 = a.last_depth[gi];
            const uint8_t li = a.last_image[gi];
            if (inimg && d0 > 0 && fabsf(transformed_d1 - d0) <= a.max_depth_delta && li != 0) {
                corres.zero_x = (int16_t)u0; corres.zero_y = (int16_t)v0;
                corres.one_x = (int16_t)x; corres.one_y = (int16_t)y;
                corres.diff = (float)ni - (float)li;
                corres.valid = 1;
                cnt += 1;
                sig += (unsigned int)kt_f2i_rz(corres.diff * corres.diff);
            }
        }
        if (PRE && !a.write_all && !candidate) continue;   // still zero from the first iteration of this level in this frame
        // one 16-byte store per DataTerm (written for every pixel, quirk A.19)
        *(int4*)&a.corres[k] = *(const int4*)&corres;
    }
    // {count, sum diff^2}: integer sums are order independent.  Wave shuffle -> workgroup -> ONE pair of {epoch, value} granules per
    // workgroup (no atomics: thousands of device-scope atomics on two addresses cost more than the kernel); the last workgroup
    // sweeps the granules (cdna_hip_programming.md G16 form R2) and, on the device path, turns them into sigmaVal for rgbStep.
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        sig += __shfl_down(sig, off, 64);
    }
    __shared__ unsigned int wsum[2][KT_RES_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wsum[0][wave] = (unsigned int)cnt; wsum[1][wave] = sig; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned int t = 0;
#pragma unroll
        for (int w = 0; w < KT_RES_THREADS / 64; ++w) t += wsum[threadIdx.x][w];
        if (a.state) {
            // device path (round 4): the pair goes out as plain words and the launch ENDS here -- no sweep.  The consumer is the next
            // launch on the stream (kt_rgb_kernel / kt_joint_kernel), whose kernel boundary makes the words visible; each of its waves adds
            // the <= 256 pairs up itself while its first loads are in flight (kt_residual_sigma): one cross-XCD polling round trip and a
            // serial tail less per Gauss-Newton iteration (5.5 -> 3.5 us).
            a.partials[threadIdx.x * KT_RES_MAX_BLOCKS + blockIdx.x] = t;
            return;
        }
        __hip_atomic_store(&a.granules[threadIdx.x * gridDim.x + blockIdx.x], ((unsigned long long)a.epoch << 32) | t, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.state || blockIdx.x != gridDim.x - 1 || threadIdx.x >= 64) return;
    // sweeping wave: lane l reads granules l, l + 64, l + 128, l + 192 of both sums -- all eight loads in flight together -- until
    // every tag carries this launch's epoch
    unsigned int tot[2] = {0, 0};
    bool ok;
    {
        const unsigned int G = gridDim.x;
        unsigned long long v[2][4];
        unsigned int spins = 0;
        const unsigned int spin_limit = *(volatile const unsigned int*)&kt_sweep_spin_limit;
        for (;;) {
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned int g = min((unsigned int)lane + 64u * q, G - 1);
                    v[which][q] = __hip_atomic_load(&a.granules[which * G + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            ok = true;
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int q = 0; q < 4; ++q) ok = ok && (unsigned int)(v[which][q] >> 32) == a.epoch;
            if (__all(ok) || ++spins > spin_limit) break;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((unsigned int)lane + 64u * q < G) tot[which] += (unsigned int)v[which][q];
    }
    for (int off = 32; off > 0; off >>= 1) {
        tot[0] += __shfl_down(tot[0], off, 64);
        tot[1] += __shfl_down(tot[1], off, 64);
    }
    const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
    if (lane == 0) {
        a.out2[0] = (int)tot[0];
        a.out2[1] = (int)tot[1];
        a.out2[2] = all_ok ? 0 : 1;
    }
}

// {count, sum diff^2} of the residual launch in front of this one, summed by every wave for itself from the workgroups' plain words
// (integer sums: any order), and RGBDOdometry.cpp:253's sigmaVal from them (quirk A.11: sqrt(count) unless sigma / count == 0)
__device__ __forceinline__ float kt_residual_sigma(const unsigned int* __restrict__ partials, int blocks, int& count_out, int& sigma_out)
{
    const unsigned int lane = threadIdx.x & 63u;
    unsigned int tot[2] = {0, 0};
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int q = 0; q < KT_RES_MAX_BLOCKS / 64; ++q) {
            const unsigned int g = lane + 64u * q;
            const unsigned int v = partials[which * KT_RES_MAX_BLOCKS + min(g, (unsigned int)blocks - 1u)];
            tot[which] += g < (unsigned int)blocks ? v : 0u;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tot[0] += __shfl_xor(tot[0], off, 64);
        tot[1] += __shfl_xor(tot[1], off, 64);
    }
    count_out = (int)tot[0];
    sigma_out = (int)tot[1];
    return __builtin_sqrtf(((float)sigma_out / (float)count_out == 0) ? 1.0f : (float)count_out);
}

extern "C" int kt_rgb_residual(kt_ctx* c, float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* last_depth,
                               const float* next_depth, const uint8_t* last_image, const uint8_t* next_image, int cols, int rows,
                               kt_dataterm* corres_img, float max_depth_delta, const float kt[3], const kt_mat33* krkinv,
                               int* sigma_sum_host, int* count_host)
{
    KT_ARG(c && dIdx && dIdy && last_depth && next_depth && last_image && next_image && corres_img && kt && krkinv && sigma_sum_host && count_host);
    KT_ARG(cols > 0 && rows > 0);
    int* out2 = (int*)&c->counters[4];
    kt_residual_args a;
    a.min_scale = min_scale; a.dIdx = dIdx; a.dIdy = dIdy; a.last_depth = last_depth; a.next_depth = next_depth;
    a.last_image = last_image; a.next_image = next_image; a.corres = corres_img; a.max_depth_delta = max_depth_delta;
    for (int k = 0; k < 3; ++k) a.kt_[k] = kt[k];
    a.krkinv = *krkinv; a.state = nullptr; a.cols = cols; a.rows = rows; a.out2 = out2;
    a.granules = kt_residual_granules(c); a.epoch = kt_next_epoch(c); a.partials = nullptr;
    a.cand = nullptr; a.write_all = 1;
    int g = kt_div_up(cols * rows, KT_RES_THREADS);
    if (g > KT_RES_MAX_BLOCKS) g = KT_RES_MAX_BLOCKS;
    hipLaunchKernelGGL(kt_residual_kernel<false>, dim3(g), dim3(KT_RES_THREADS), 0, c->stream, a);
    KT_LAUNCH_CHECK();
    KT_HIP(hipMemcpyAsync(c->int_out_host, out2, 3 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    if (c->int_out_host[2]) { (void)kt_refill_granules(c); kt_set_error("computeRgbResidual: inter-workgroup hand-off timed out"); return KT_ERR_STATE; }
    *count_host = c->int_out_host[0];
    *sigma_sum_host = c->int_out_host[1];
    return KT_OK;
}

int kt_rgb_residual_device(kt_ctx* c, kt_track_state* state, float min_scale, const int16_t* dIdx, const int16_t* dIdy,
                           const float* last_depth, const float* next_depth, const uint8_t* last_image, const uint8_t* next_image,
                           int cols, int rows, kt_dataterm* corres_img, float max_depth_delta, const uint8_t* cand, int write_all)
{
    kt_residual_args a;
    a.min_scale = min_scale; a.dIdx = dIdx; a.dIdy = dIdy; a.last_depth = last_depth; a.next_depth = next_depth;
    a.last_image = last_image; a.next_image = next_image; a.corres = corres_img; a.max_depth_delta = max_depth_delta;
    a.state = state; a.cols = cols; a.rows = rows; a.out2 = (int*)&c->counters[4];
    a.granules = nullptr; a.epoch = 0; a.partials = kt_residual_partials(c);
    a.cand = cand; a.write_all = write_all;
    const int g = kt_residual_blocks(cols, rows);
    if (cand) hipLaunchKernelGGL(kt_residual_kernel<true>, dim3(g), dim3(KT_RES_THREADS), 0, c->stream, a);
    else hipLaunchKernelGGL(kt_residual_kernel<false>, dim3(g), dim3(KT_RES_THREADS), 0, c->stream, a);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// ------------------------------------------------------------------------------------------------
// a9  rgbStep -> rgbKernel + reduceSum                reduce.cu:423-607
// ------------------------------------------------------------------------------------------------
struct kt_rgb_args {
    const kt_dataterm* corres;
    float sigma;
    const float* cloud;
    float fx, fy;
    const int16_t* dIdx; const int16_t* dIdy;
    float sobel_scale;
    int cols, rows;
    kt_track_state* state;
    unsigned long long* granules;
    float* out29;
    int mode;              // KT_MODE_HOST, KT_MODE_RGB_SOLVE, KT_MODE_JOINT_SOLVE
    kt_level_k next_k;     // intrinsics of the level the NEXT iteration runs at (for K R K^-1, K t)
    const unsigned int* res_partials; int res_blocks;   // device path: the residual launch's per-workgroup {count, sum diff^2} words
};

struct kt_rgb_row {
    const kt_rgb_args& a;
    float sigma;
    // kt_joint_level_kernel: the DataTerms were stored by THIS workgroup a moment ago (same launch): read them past the compute unit's vector
    // cache (workgroup-scope loads; the stores went through it to the XCD's L2, a line cached from the previous iteration's read would be stale)
    bool own_terms = false;
    // RGBReduction::getProducts, reduce.cu:441-486.  Branch-free like kt_icp_row (the reference's early return for an invalid DataTerm
    // becomes a predicate, the gather indices of an invalid pixel are clamped to 0) and staged: fetch_term -> fetch_taps -> finish.
    struct taps { float X, Y, Z, gx, gy; };
    __device__ __forceinline__ kt_dataterm fetch_term(int i) const
    {
        if (own_terms) {
            const unsigned long long* q = (const unsigned long long*)&a.corres[i];
            unsigned long long w[2];
            w[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            w[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return *(const kt_dataterm*)&w[0];
        }
        const int4 raw = *(const int4*)&a.corres[i];
        return *(const kt_dataterm*)&raw;
    }
    __device__ __forceinline__ void fetch_taps(const kt_dataterm& c, taps& t) const
    {
        const bool valid = c.valid != 0;
        const int zi = valid ? c.zero_y * a.cols + c.zero_x : 0, oi = valid ? c.one_y * a.cols + c.one_x : 0;
        const float* cp = &a.cloud[3 * zi];
        t.X = cp[0]; t.Y = cp[1]; t.Z = cp[2];
        t.gx = (float)a.dIdx[oi]; t.gy = (float)a.dIdy[oi];
    }
    __device__ __forceinline__ bool finish(const kt_dataterm& c, const taps& t, float (&row)[7]) const
    {
        const float flt_eps = 1.19209290E-07F;
        const bool valid = c.valid != 0;
        float w = sigma + fabsf(c.diff);
        w = w > flt_eps ? 1.0f / w : 1.0f;
        if (sigma == -1) w = 1;
        const float r6 = -w * c.diff;
        const float X = t.X, Y = t.Y, Z = t.Z;
        const float invz = (float)(1.0 / (double)Z);
        const float dI_dx_val = w * a.sobel_scale * t.gx;
        const float dI_dy_val = w * a.sobel_scale * t.gy;
        const float v0 = dI_dx_val * a.fx * invz;
        const float v1 = dI_dy_val * a.fy * invz;
        const float v2 = -__builtin_fmaf(v0, X, v1 * Y) * invz;
        row[0] = valid ? v0 : 0.0f; row[1] = valid ? v1 : 0.0f; row[2] = valid ? v2 : 0.0f;
        row[3] = valid ? __builtin_fmaf(Y, v2, -(Z * v1)) : 0.0f;   // -Z*v1 + Y*v2 is canonicalised to Y*v2 - Z*v1 before contraction (oracle/_ref)
        row[4] = valid ? __builtin_fmaf(Z, v0, -(X * v2)) : 0.0f;
        row[5] = valid ? __builtin_fmaf(X, v1, -(Y * v0)) : 0.0f;   // likewise: X*v1 - Y*v0
        row[6] = valid ? r6 : 0.0f;
        return valid;
    }
    __device__ __forceinline__ bool operator()(int i, float (&row)[7]) const
    {
        const kt_dataterm c = fetch_term(i);
        taps t;
        fetch_taps(c, t);
        return finish(c, t, row);
    }
};

__global__ __launch_bounds__(KT_RED_THREADS) void kt_rgb_kernel(const kt_rgb_args a)
{
    int res_count = 0, res_sigma = 0;
    const kt_rgb_row fn{a, a.state ? kt_residual_sigma(a.res_partials, a.res_blocks, res_count, res_sigma) : a.sigma};
    __shared__ float total[KT_RED_SLOTS];
    kt_pose_stage ps;
    auto pre = [&]() { if (a.mode != KT_MODE_HOST) ps.fetch(a.state); };
    if (!kt_reduce29(fn, a.cols * a.rows, a.granules, total, pre)) return;
    if (a.mode == KT_MODE_HOST) {
        if (threadIdx.x < KT_RED_SLOTS && (threadIdx.x < 29 || threadIdx.x == KT_RED_SLOTS - 1)) a.out29[threadIdx.x] = total[threadIdx.x];
    } else {
        __shared__ double sys[KT_SYS_DOUBLES], pose_d[KT_POSE_STAGE_DOUBLES], tail_work[KT_TAIL_WORK_DOUBLES];
        __shared__ float pose_f[KT_POSE_STAGE_FLOATS];
        ps.park(pose_d, pose_f);
        if (threadIdx.x < 42) {
            const int slot = kt_sys_slot(threadIdx.x);
            double v = (double)total[slot];
            if (a.mode == KT_MODE_JOINT_SOLVE) {
                // RGBDOdometry.cpp:316-321: A = A_rgbd + w*w*A_icp, b = b_rgbd + w*b_icp, w = 10
                const double w = 10, vi = (double)a.state->icp29[slot];
                v = threadIdx.x < 36 ? v + w * w * vi : v + w * vi;
            }
            sys[threadIdx.x] = v;
        }
        __syncthreads();
        if (threadIdx.x == 64) {
            if (total[KT_RED_SLOTS - 1] != 0.0f) a.state->handoff_timeout = 1;
            a.state->sigma_val = fn.sigma; a.state->rgb_count = res_count; a.state->rgb_sigma = res_sigma;   // (kept for observers)
        }
        if (threadIdx.x < 64) {
            kt_solve_and_update_wave(a.state, sys, pose_d, pose_f, tail_work);
            __builtin_amdgcn_wave_barrier();
            if (threadIdx.x == 0) kt_compute_krk(pose_d, a.next_k, a.state->krkinv, a.state->kt);
        }
    }
}

// Joint RGB-D + ICP iteration in ONE launch (RGBDOdometry.cpp:262-321): both 29-sum reductions are published by every workgroup, the
// last one sweeps both, combines A = A_rgbd + w^2 A_icp, b = b_rgbd + w b_icp (w = 10), solves and updates the pose.  Replaces the
// kt_icp_kernel(KT_MODE_ICP_STASH) + kt_rgb_kernel(KT_MODE_JOINT_SOLVE) pair: one kernel boundary and one sweep less per iteration.
__global__ __launch_bounds__(KT_RED_THREADS) void kt_joint_kernel(const kt_icp_args ai, const kt_rgb_args ar)
{
    kt_icp_row fi(ai);
    for (int k = 0; k < 9; ++k) { fi.Rcurr.m[k] = ai.state->Rcurr[k]; fi.Rprev_inv.m[k] = ai.state->Rprev_inv[k]; }
    fi.tcurr = {ai.state->tcurr[0], ai.state->tcurr[1], ai.state->tcurr[2]};
    fi.tprev = {ai.state->tprev[0], ai.state->tprev[1], ai.state->tprev[2]};
    int res_count = 0, res_sigma = 0;
    const kt_rgb_row fr{ar, kt_residual_sigma(ar.res_partials, ar.res_blocks, res_count, res_sigma)};
    __shared__ kt_rows_t rows_icp[KT_KBATCH], rows_rgb[KT_KBATCH];   // 2 x 40 KB
    __shared__ float total_icp[KT_RED_SLOTS], total[KT_RED_SLOTS];
    if (kt_red_publishes()) kt_reduce29_publish2(fi, fr, ai.cols * ai.rows, ai.granules, ar.granules, rows_icp, rows_rgb);
    if (!kt_red_sweeps()) return;
    kt_pose_stage ps;
    ps.fetch(ar.state);
    {
        unsigned long long* const gs[2] = {ai.granules, ar.granules};
        float* const ts[2] = {total_icp, total};   // the time-out flag lands in total[31]
        kt_reduce29_sweep_n<2>(gs, ts);
    }
    __shared__ double sys[KT_SYS_DOUBLES], pose_d[KT_POSE_STAGE_DOUBLES], tail_work[KT_TAIL_WORK_DOUBLES];
    __shared__ float pose_f[KT_POSE_STAGE_FLOATS];
    ps.park(pose_d, pose_f);
    if (threadIdx.x < 42) {   // RGBDOdometry.cpp:316-321: A = A_rgbd + w*w*A_icp, b = b_rgbd + w*b_icp, w = 10
        const int slot = kt_sys_slot(threadIdx.x);
        const double w = 10, v = (double)total[slot], vi = (double)total_icp[slot];
        sys[threadIdx.x] = threadIdx.x < 36 ? v + w * w * vi : v + w * vi;
    }
    __syncthreads();
    if (threadIdx.x == 64) {
        if (total[KT_RED_SLOTS - 1] != 0.0f) ar.state->handoff_timeout = 1;
        ar.state->sigma_val = fr.sigma; ar.state->rgb_count = res_count; ar.state->rgb_sigma = res_sigma;   // (kept for observers)
    }
    if (threadIdx.x < 64) {
        kt_solve_and_update_wave(ar.state, sys, pose_d, pose_f, tail_work);
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) kt_compute_krk(pose_d, ar.next_k, ar.state->krkinv, ar.state->kt);
#ifdef KT_ICP_TIMING
        if (threadIdx.x == 0) { const unsigned long long t7 = wall_clock64(); for (int q = 0; q < 7; ++q) ar.state->icp29[q] = (float)(kt_ts[q + 0] - kt_ts[0]); ar.state->icp29[7] = (float)(t7 - kt_ts[0]); }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// The -ri iterations of ONE pyramid level in one launch (round 6).  Until now an iteration was two launches because the
// Jacobian rows of rgbStep need the GRID-WIDE correspondence count of computeRgbResidual (sigmaVal, RGBDOdometry.cpp:253) before a single
// pixel can be weighed.  Here the 256 workgroups stay resident, as in kt_icp_level_kernel, and an iteration is
//   R  every workgroup runs residualKernel's body (reduce.cu:718-760) for ITS OWN pixels -- the pixels its 32 virtual threads will reduce, so
//      the DataTerms it writes (every pixel, quirk A.19) are read back by nobody else -- and publishes its {count, sum diff^2} as two tagged
//      granules; every workgroup then collects the 512 granules (integer sums: any order) and forms sigmaVal for itself;
//   J  kt_joint_kernel's pass: the ICP row and the RGB-D row of every pixel, both 29-sum reductions in the reference's order, two granule sets;
//   T  the last workgroup sweeps both sets, combines A = A_rgbd + 100 A_icp, b = b_rgbd + 10 b_icp, solves, updates the pose and K R K^-1 / K t
//      (kt_compute_krk) and publishes BOTH as 24 tagged granules, which the other workgroups poll.
// Two in-kernel exchanges (0.7-1 us each) instead of two kernel boundaries and their first loads (2 x ~2.1 us), and the DataTerm image no longer
// crosses the chip between two launches.  Arithmetic: kt_residual_kernel's and kt_joint_kernel's, operation for operation (tests: config 3
// byte-equal to the oracle, tests/test_gpu_tracker.py level == stepwise).  Waits, abort and fallback: kt_icp_level_kernel's.
// ------------------------------------------------------------------------------------------------
#define KT_KRK_GRAN 16   // pose_gran[16 .. 28): K R K^-1 (9) and K t (3) of the iteration, tagged like the pose granules [0, 12)
struct kt_joint_level_extra {
    int n_iter; unsigned int seq0;
    unsigned long long* pose_gran; unsigned long long* level_gran;   // level_gran: 4 sets: (it & 1) * 2 + {0: ICP, 1: RGB-D}
    unsigned long long* res_gran;                                    // [2][KT_RED_BLOCKS] {seq << 32 | value}: count, sum diff^2
    kt_level_k k_level, k_next;   // K of this level (iterations that stay on it) and of the level the iteration behind the launch runs at
    unsigned int fault;
};

__global__ __launch_bounds__(KT_RED_THREADS) void kt_joint_level_kernel(const kt_icp_args ai, const kt_rgb_args ar, const kt_residual_args rr, const kt_joint_level_extra x)
{
    if (x.fault && blockIdx.x == 0) return;   // test hook (kt_debug_handoff_fault)
    if (ai.state->handoff_timeout) return;    // an earlier launch of this frame gave up (the state is uploaded clean at the start of every -ri frame)
    __shared__ kt_rows_t rows_icp[KT_KBATCH], rows_rgb[KT_KBATCH];   // 2 x 40 KB
    __shared__ float total_icp[KT_RED_SLOTS], total[KT_RED_SLOTS];
    __shared__ double sys[KT_SYS_DOUBLES], pose_d[KT_POSE_STAGE_DOUBLES], tail_work[KT_TAIL_WORK_DOUBLES];
    __shared__ float pose_f[KT_POSE_STAGE_FLOATS];
    __shared__ float s_pose[25];            // pose (12), K R K^-1 (9), K t (3) of the iteration about to run; [24] != 0: they never came
    __shared__ unsigned int s_res[3];       // count, sum diff^2 of the iteration; [2] != 0: the partial sums never came
    __shared__ unsigned int wsum[2][KT_RED_THREADS / 64];
    const bool sweeper = kt_red_sweeps();
    kt_icp_row fi(ai);
    for (int k = 0; k < 9; ++k) { fi.Rcurr.m[k] = ai.state->Rcurr[k]; fi.Rprev_inv.m[k] = ai.state->Rprev_inv[k]; }
    fi.tcurr = {ai.state->tcurr[0], ai.state->tcurr[1], ai.state->tcurr[2]};
    fi.tprev = {ai.state->tprev[0], ai.state->tprev[1], ai.state->tprev[2]};
    float K[9], kt[3];
    for (int q = 0; q < 9; ++q) K[q] = ai.state->krkinv[q];
    for (int q = 0; q < 3; ++q) kt[q] = ai.state->kt[q];
    kt_rgb_row fr{ar, 0.0f};
    fr.own_terms = true;
    if (sweeper) {   // the solving workgroup's carry (kt_pose_stage's layout)
        if (threadIdx.x < 16) pose_d[threadIdx.x] = ai.state->resultRt[threadIdx.x];
        else if (threadIdx.x < 25) pose_f[threadIdx.x - 16] = ai.state->Rprev[threadIdx.x - 16];
        else if (threadIdx.x < 28) pose_f[threadIdx.x - 16] = ai.state->tprev[threadIdx.x - 25];
    }
    if (threadIdx.x == 24) s_pose[24] = 0.0f;
    const int cols = ai.cols, rows = ai.rows, n = cols * rows;
    const int t0 = blockIdx.x * 32, nk_blk = (n - t0 + KT_VT_TOTAL - 1) / KT_VT_TOTAL;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    for (int it = 0; it < x.n_iter; ++it) {
        const unsigned int seq = x.seq0 + (unsigned int)it;
        if (it > 0) {
            if (!sweeper && threadIdx.x < 64) {
                // pose + K R K^-1 + K t of iteration it - 1: 24 granules in two lines, and the abort granule (kt_icp_level_kernel's wait)
                const unsigned int want = seq - 1u;
                const unsigned int spin_limit = min(*(volatile const unsigned int*)&kt_sweep_spin_limit, 1u << 20);
                const unsigned int tick_limit = 2u * *(volatile const unsigned int*)&kt_wait_limit_ticks;
                unsigned long long tick0 = 0, g = 0;
                unsigned int spins = 0;
                const bool mine = lane < 13 || (lane >= KT_KRK_GRAN && lane < KT_KRK_GRAN + 12);
                const int gi = lane == 12 ? KT_POSE_ABORT : lane;
                bool ok, gone;
                for (;;) {
                    if (mine) g = __hip_atomic_load(&x.pose_gran[gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int ahead = (int)((unsigned int)(g >> 32) - want);
                    ok = !mine || lane == 12 || ahead == 0;
                    gone = lane == 12 && ahead >= 0 && ahead < x.n_iter;
                    if (__all(ok) || __any(gone) || ++spins > spin_limit) break;
                    if ((spins & 15u) == 0) {
                        const unsigned long long now = kt_ticks();
                        if (spins == 16u) tick0 = now;
                        else if (now - tick0 > tick_limit) break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const bool got = __all(ok) && !__any(gone);
                if (lane < 12) s_pose[lane] = __uint_as_float((unsigned int)g);
                if (lane >= KT_KRK_GRAN && lane < KT_KRK_GRAN + 12) s_pose[12 + lane - KT_KRK_GRAN] = __uint_as_float((unsigned int)g);
                if (lane == 12) s_pose[24] = got ? 0.0f : 1.0f;
            }
            __syncthreads();   // (the solving workgroup: its tail wrote s_pose)
            if (s_pose[24] != 0.0f) return;   // workgroup-uniform: no pose; the sweep of this iteration cannot complete and reports it
            for (int k = 0; k < 9; ++k) fi.Rcurr.m[k] = uni(s_pose[k]);
            fi.tcurr = {uni(s_pose[9]), uni(s_pose[10]), uni(s_pose[11])};
            for (int k = 0; k < 9; ++k) K[k] = uni(s_pose[12 + k]);
            for (int k = 0; k < 3; ++k) kt[k] = uni(s_pose[21 + k]);
        }
        // ---- R: computeRgbResidual for this workgroup's own pixels (the pixels of its 32 virtual threads, batch by batch like the reduction)
        {
            int cnt = 0;
            unsigned int sig = 0;   // wraps modulo 2^32 like the reference's int sum
            for (int kb = 0; kb < nk_blk; kb += KT_KBATCH) {
                const int kcount = min(KT_KBATCH, nk_blk - kb);
                for (int p = (int)threadIdx.x; p < kcount * 32; p += KT_RED_THREADS) {
                    const int i = t0 + (p & 31) + (kb + (p >> 5)) * KT_VT_TOTAL;
                    const bool inside = i < n;
                    const int k = inside ? i : n - 1;
                    const float d1 = rr.next_depth[k];
                    const uint8_t ni = rr.next_image[k];
                    const bool candidate = inside && rr.cand[k] != 0;
                    kt_dataterm corres;
                    corres.zero_x = corres.zero_y = corres.one_x = corres.one_y = 0; corres.diff = 0.f; corres.valid = 0;
                    corres.pad[0] = corres.pad[1] = corres.pad[2] = 0;
                    if (__builtin_amdgcn_ballot_w64(candidate) != 0) {   // wave-uniform: no candidate among these 64 pixels, no arithmetic
                        if (kt_residual_pixel(rr, K, kt, k, cols, rows, candidate, d1, ni, corres)) {
                            cnt += 1;
                            sig += (unsigned int)kt_f2i_rz(corres.diff * corres.diff);
                        }
                    }
                    // the first iteration of a level writes every DataTerm (quirk A.19), later ones only the candidates: the others are still zero
                    if (inside && (it == 0 ? rr.write_all != 0 || candidate : candidate)) *(int4*)&rr.corres[k] = *(const int4*)&corres;
                }
            }
            for (int off = 32; off > 0; off >>= 1) { cnt += __shfl_down(cnt, off, 64); sig += __shfl_down(sig, off, 64); }
            if (lane == 0) { wsum[0][wave] = (unsigned int)cnt; wsum[1][wave] = sig; }
            // the DataTerm stores have reached the XCD's L2 (write-through) before any wave of the workgroup reads them back in J
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x < 2) {
                unsigned int t = 0;
#pragma unroll
                for (int w = 0; w < KT_RED_THREADS / 64; ++w) t += wsum[threadIdx.x][w];
                __hip_atomic_store(&x.res_gran[threadIdx.x * KT_RED_BLOCKS + blockIdx.x], ((unsigned long long)seq << 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // every workgroup collects all 2 x 256 partial sums (wave 1: the sums are integers, any order will do)
            if (wave == 1) {
                unsigned long long v[2][4];
                const unsigned int spin_limit = min(*(volatile const unsigned int*)&kt_sweep_spin_limit, 1u << 20);
                const unsigned int tick_limit = 2u * *(volatile const unsigned int*)&kt_wait_limit_ticks;
                unsigned long long tick0 = 0;
                unsigned int spins = 0;
                bool ok;
                for (;;) {
#pragma unroll
                    for (int which = 0; which < 2; ++which)
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[which][q] = __hip_atomic_load(&x.res_gran[which * KT_RED_BLOCKS + lane + 64 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = true;
#pragma unroll
                    for (int which = 0; which < 2; ++which)
#pragma unroll
                        for (int q = 0; q < 4; ++q) ok = ok && (unsigned int)(v[which][q] >> 32) == seq;
                    if (__all(ok) || ++spins > spin_limit) break;
                    if ((spins & 15u) == 0) {
                        const unsigned long long now = kt_ticks();
                        if (spins == 16u) tick0 = now;
                        else if (now - tick0 > tick_limit) break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                unsigned int tot[2] = {0, 0};
#pragma unroll
                for (int which = 0; which < 2; ++which)
#pragma unroll
                    for (int q = 0; q < 4; ++q) tot[which] += (unsigned int)v[which][q];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { tot[0] += __shfl_xor(tot[0], off, 64); tot[1] += __shfl_xor(tot[1], off, 64); }
                if (lane == 0) { s_res[0] = tot[0]; s_res[1] = tot[1]; s_res[2] = __all(ok) ? 0u : 1u; }
            }
            __syncthreads();
            if (s_res[2] != 0u && !sweeper) return;   // (the sweeper goes on: its sweep reports the time-out)
        }
        const int res_count = (int)s_res[0], res_sigma = (int)s_res[1];
        fr.sigma = __builtin_sqrtf(((float)res_sigma / (float)res_count == 0) ? 1.0f : (float)res_count);   // RGBDOdometry.cpp:253 (kt_residual_sigma)
        // ---- J: both rows of every pixel, both reductions (kt_joint_kernel's pass)
        unsigned long long* const gran_icp = x.level_gran + (size_t)((it & 1) * 2) * KT_LEVEL_SET_STRIDE;
        unsigned long long* const gran_rgb = gran_icp + KT_LEVEL_SET_STRIDE;
        kt_reduce29_publish2(fi, fr, n, gran_icp, gran_rgb, rows_icp, rows_rgb);
        if (!sweeper) continue;
        // ---- T
        {
            unsigned long long* const gs[2] = {gran_icp, gran_rgb};
            float* const ts[2] = {total_icp, total};   // the time-out flag lands in total[31]
            kt_reduce29_sweep_n<2>(gs, ts, 1u);
        }
        const bool timed_out = total[KT_RED_SLOTS - 1] != 0.0f || s_res[2] != 0u;
        if (threadIdx.x < 42) {   // RGBDOdometry.cpp:316-321: A = A_rgbd + w*w*A_icp, b = b_rgbd + w*b_icp, w = 10
            const int slot = kt_sys_slot(threadIdx.x);
            const double w = 10, v = (double)total[slot], vi = (double)total_icp[slot];
            sys[threadIdx.x] = threadIdx.x < 36 ? v + w * w * vi : v + w * vi;
        }
        __syncthreads();
        if (threadIdx.x == 64) {
            if (timed_out) {
                ar.state->handoff_timeout = 1;
                __hip_atomic_store(&x.pose_gran[KT_POSE_ABORT], (unsigned long long)seq << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ar.state->sigma_val = fr.sigma; ar.state->rgb_count = res_count; ar.state->rgb_sigma = res_sigma;   // (kept for observers)
        }
        if (timed_out) return;
        if (threadIdx.x < 64) {
            kt_solve_and_update_wave(ar.state, sys, pose_d, pose_f, tail_work, nullptr, 0, s_pose);
            __builtin_amdgcn_wave_barrier();
            // K R K^-1, K t for the iteration behind this one: at this level's K, or -- behind the launch's last iteration -- the next level's
            if (threadIdx.x == 0) {
                float krk[9], ktn[3];
                kt_compute_krk(pose_d, it + 1 < x.n_iter ? x.k_level : x.k_next, krk, ktn);
                for (int q = 0; q < 9; ++q) { ar.state->krkinv[q] = krk[q]; s_pose[12 + q] = krk[q]; }
                for (int q = 0; q < 3; ++q) { ar.state->kt[q] = ktn[q]; s_pose[21 + q] = ktn[q]; }
            }
            __builtin_amdgcn_wave_barrier();
            // the hand-back of both granule sets has completed in waves 1..15 (they arrive at the barrier once their stores have): now the
            // pose and the warp may be seen
            __builtin_amdgcn_s_barrier();
            if (lane < 24) {
                const float v = s_pose[lane];
                __hip_atomic_store(&x.pose_gran[lane < 12 ? lane : KT_KRK_GRAN + lane - 12], ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// n_iter joint RGB-D + ICP iterations of one pyramid level in one launch (kt_joint_level_kernel).  k_level: this level's intrinsics; k_next: those of
// the level the iteration BEHIND this launch runs at (the warp K R K^-1, K t is formed by the iteration in front of the one that uses it).
int kt_joint_level_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr, const float* vmap_g_prev,
                          const float* nmap_g_prev, float dist_thres, float angle_thres, kt_dataterm* corres_img, const float* cloud, const int16_t* dIdx,
                          const int16_t* dIdy, float sobel_scale, float min_scale, const float* last_depth, const float* next_depth, const uint8_t* last_image,
                          const uint8_t* next_image, float max_depth_delta, const uint8_t* cand, int cols, int rows, int n_iter, const kt_level_k* k_level,
                          const kt_level_k* k_next)
{
    if (n_iter <= 0) return KT_OK;
    KT_ARG(cand && k_level && k_next);
    kt_icp_args a;
    memset(&a, 0, sizeof(a));
    a.vmap_curr = vmap_curr; a.nmap_curr = nmap_curr; a.vmap_g_prev = vmap_g_prev; a.nmap_g_prev = nmap_g_prev;
    a.intr = *intr; a.cols = cols; a.rows = rows; kt_icp_set_thresholds(a, dist_thres, angle_thres);
    a.state = state; a.out29 = nullptr; a.mode = KT_MODE_ICP_STASH;
    kt_rgb_args r;
    memset(&r, 0, sizeof(r));
    r.corres = corres_img; r.sigma = 0.f; r.cloud = cloud; r.fx = intr->fx; r.fy = intr->fy; r.dIdx = dIdx; r.dIdy = dIdy;
    r.sobel_scale = sobel_scale; r.cols = cols; r.rows = rows; r.state = state; r.mode = KT_MODE_JOINT_SOLVE; r.next_k = *k_level;
    kt_residual_args rr;
    memset(&rr, 0, sizeof(rr));
    rr.min_scale = min_scale; rr.dIdx = dIdx; rr.dIdy = dIdy; rr.last_depth = last_depth; rr.next_depth = next_depth;
    rr.last_image = last_image; rr.next_image = next_image; rr.corres = corres_img; rr.max_depth_delta = max_depth_delta;
    rr.state = state; rr.cols = cols; rr.rows = rows; rr.cand = cand; rr.write_all = 1;
    kt_joint_level_extra x;
    x.n_iter = n_iter;
    if (c->odo_seq > 0xfffff000u) c->odo_seq = 0;   // the tags never reach the buffer's fill pattern (all ones)
    x.seq0 = c->odo_seq + 1u;
    c->odo_seq += (unsigned int)n_iter;
    x.pose_gran = c->pose_gran;
    x.level_gran = (unsigned long long*)c->red_partials + KT_LEVEL_SET_BASE;
    x.res_gran = kt_residual_granules(c) + 2048;
    x.k_level = *k_level; x.k_next = *k_next;
    x.fault = 0;
    if (c->fault_skip > 0) --c->fault_skip;
    else if (c->fault_count > 0) { --c->fault_count; x.fault = 1; }
    hipLaunchKernelGGL(kt_joint_level_kernel, dim3(KT_RED_GRID), dim3(KT_RED_THREADS), 0, c->stream, a, r, rr, x);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

int kt_joint_step_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr,
                         const float* vmap_g_prev, const float* nmap_g_prev, float dist_thres, float angle_thres,
                         const kt_dataterm* corres_img, const float* cloud, const int16_t* dIdx, const int16_t* dIdy, float sobel_scale,
                         int cols, int rows, const kt_level_k* next_k)
{
    kt_icp_args a;
    a.vmap_curr = vmap_curr; a.nmap_curr = nmap_curr; a.vmap_g_prev = vmap_g_prev; a.nmap_g_prev = nmap_g_prev;
    a.intr = *intr; a.cols = cols; a.rows = rows; kt_icp_set_thresholds(a, dist_thres, angle_thres);
    a.state = state; a.out29 = nullptr; a.mode = KT_MODE_ICP_STASH; a.first = 0; a.keep29 = 0;
    a.granules = kt_second_granules(c);
    a.fault = 0;
    kt_rgb_args r;
    r.corres = corres_img; r.sigma = 0.f; r.cloud = cloud; r.fx = intr->fx; r.fy = intr->fy; r.dIdx = dIdx; r.dIdy = dIdy;
    r.sobel_scale = sobel_scale; r.cols = cols; r.rows = rows; r.state = state;
    r.granules = (unsigned long long*)c->red_partials; r.out29 = nullptr; r.mode = KT_MODE_JOINT_SOLVE;
    r.next_k = *next_k;
    r.res_partials = kt_residual_partials(c); r.res_blocks = kt_residual_blocks(cols, rows);
    hipLaunchKernelGGL(kt_joint_kernel, dim3(KT_RED_GRID), dim3(KT_RED_THREADS), 0, c->stream, a, r);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

extern "C" int kt_rgb_step(kt_ctx* c, const kt_dataterm* corres_img, float sigma, const float* cloud, float fx, float fy,
                           const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows, float* A_host,
                           float* b_host)
{
    KT_ARG(c && corres_img && cloud && dIdx && dIdy && A_host && b_host && cols > 0 && rows > 0);
    kt_rgb_args a;
    a.corres = corres_img; a.sigma = sigma; a.cloud = cloud; a.fx = fx; a.fy = fy; a.dIdx = dIdx; a.dIdy = dIdy;
    a.sobel_scale = sobel_scale; a.cols = cols; a.rows = rows; a.state = nullptr;
    a.granules = (unsigned long long*)c->red_partials; a.out29 = c->red_out; a.mode = KT_MODE_HOST;
    a.res_partials = nullptr; a.res_blocks = 0;
    hipLaunchKernelGGL(kt_rgb_kernel, dim3(KT_RED_GRID), dim3(KT_RED_THREADS), 0, c->stream, a);
    KT_LAUNCH_CHECK();
    KT_HIP(hipMemcpyAsync(c->red_out_host, c->red_out, sizeof(float) * KT_RED_SLOTS, hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    if (c->red_out_host[KT_RED_SLOTS - 1] != 0.0f) { (void)kt_refill_granules(c); kt_set_error("rgbStep: inter-workgroup hand-off timed out"); return KT_ERR_STATE; }
    kt_unpack29_host(c->red_out_host, A_host, b_host, nullptr);
    return KT_OK;
}

int kt_rgb_step_device(kt_ctx* c, kt_track_state* state, const kt_dataterm* corres_img, const float* cloud, float fx, float fy,
                       const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows, int mode,
                       const kt_level_k* next_k)
{
    kt_rgb_args a;
    a.corres = corres_img; a.sigma = 0.f; a.cloud = cloud; a.fx = fx; a.fy = fy; a.dIdx = dIdx; a.dIdy = dIdy;
    a.sobel_scale = sobel_scale; a.cols = cols; a.rows = rows; a.state = state;
    a.granules = (unsigned long long*)c->red_partials; a.out29 = nullptr; a.mode = mode;
    a.next_k = *next_k;
    a.res_partials = kt_residual_partials(c); a.res_blocks = kt_residual_blocks(cols, rows);
    hipLaunchKernelGGL(kt_rgb_kernel, dim3(KT_RED_GRID), dim3(KT_RED_THREADS), 0, c->stream, a);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// ------------------------------------------------------------------------------------------------
// test hook: the serial tail (kt_solve_and_update, one thread) and the lane-parallel one (kt_solve_and_update_wave) on the same systems
// ------------------------------------------------------------------------------------------------
struct kt_solve_case {
    float packed[32];    // the 29 sums in reduce.cu:401-418 order (27 of them the upper triangle of [A | b])
    float packed2[32];   // joint != 0: the ICP sums of the joint solve (A = A_rgbd + 100 A_icp, b = b_rgbd + 10 b_icp)
    double resultRt[16];
    float posef[12];     // Rprev[9], tprev[3]
    int joint, pad[3];
};

__global__ __launch_bounds__(64) void kt_solve_check_kernel(const kt_solve_case* __restrict__ cases, kt_track_state* serial_out, kt_track_state* wave_out)
{
    __shared__ double sys[KT_SYS_DOUBLES], pose_d[KT_POSE_STAGE_DOUBLES], tail_work[KT_TAIL_WORK_DOUBLES];
    __shared__ float pose_f[KT_POSE_STAGE_FLOATS];
    const kt_solve_case& cs = cases[blockIdx.x];
    if (threadIdx.x < 42) {
        const int slot = kt_sys_slot(threadIdx.x);
        double v = (double)cs.packed[slot];
        if (cs.joint) {
            const double w = 10, vi = (double)cs.packed2[slot];
            v = threadIdx.x < 36 ? v + w * w * vi : v + w * vi;
        }
        sys[threadIdx.x] = v;
    }
    if (threadIdx.x < 16) pose_d[threadIdx.x] = cs.resultRt[threadIdx.x];
    if (threadIdx.x < 12) pose_f[threadIdx.x] = cs.posef[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        kt_pose_regs pr;
        pr.load(pose_d, pose_f);
        kt_solve_and_update(&serial_out[blockIdx.x], pr, sys);
    }
    __syncthreads();
    kt_solve_and_update_wave(&wave_out[blockIdx.x], sys, pose_d, pose_f, tail_work);
}

extern "C" int kt_debug_solve_check(kt_ctx* c, int n, const void* cases_host, void* serial_out_host, void* wave_out_host, int layout_out[5])
{
    KT_ARG(c && layout_out);
    layout_out[0] = (int)sizeof(kt_track_state); layout_out[1] = (int)offsetof(kt_track_state, resultRt);
    layout_out[2] = (int)offsetof(kt_track_state, Rcurr); layout_out[3] = (int)offsetof(kt_track_state, tcurr);
    layout_out[4] = (int)sizeof(kt_solve_case);
    if (n <= 0) return KT_OK;
    KT_ARG(cases_host && serial_out_host && wave_out_host);
    kt_solve_case* cases = nullptr;
    kt_track_state* out = nullptr;
    KT_HIP(hipMalloc((void**)&cases, sizeof(kt_solve_case) * (size_t)n));
    if (hipMalloc((void**)&out, sizeof(kt_track_state) * 2 * (size_t)n) != hipSuccess) { (void)hipFree(cases); kt_set_error("kt_debug_solve_check: out of memory"); return KT_ERR_HIP; }
    int status = KT_OK;
    if (hipMemcpyAsync(cases, cases_host, sizeof(kt_solve_case) * (size_t)n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemsetAsync(out, 0, sizeof(kt_track_state) * 2 * (size_t)n, c->stream) != hipSuccess) status = KT_ERR_HIP;
    if (status == KT_OK) {
        hipLaunchKernelGGL(kt_solve_check_kernel, dim3(n), dim3(64), 0, c->stream, cases, out, out + n);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(serial_out_host, out, sizeof(kt_track_state) * (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(wave_out_host, out + n, sizeof(kt_track_state) * (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) status = KT_ERR_HIP;
    }
    (void)hipFree(cases);
    (void)hipFree(out);
    if (status != KT_OK) kt_set_error("kt_debug_solve_check: HIP error");
    return status;
}

// analysis hook (builds with -DKT_ICP_TIMING): when every workgroup of the LAST reduction launch entered its pixel loop, left it and had
// published its granules (100 MHz ticks; scripts/icp_timing.py)
extern "C" int kt_debug_icp_wg_times(kt_ctx* c, unsigned long long* out768_host)
{
    KT_ARG(c && out768_host);
#ifdef KT_ICP_TIMING
    KT_HIP(hipStreamSynchronize(c->stream));
    KT_HIP(hipMemcpyFromSymbol(out768_host, HIP_SYMBOL(kt_wg_ts), sizeof(unsigned long long) * 768));
    return KT_OK;
#else
    kt_set_error("kt_debug_icp_wg_times: the library was not built with -DKT_ICP_TIMING");
    return KT_ERR_STATE;
#endif
}


// ---- test hook: a reduction launch that loses a publisher ---------------------------------------------------------------------------
// After `skip` more ICP reduction launches on this context, `count` launches run without workgroup 0 (it returns at once), so their
// sweeps give up after spin_limit looks (0 = leave the limit alone; the product's is 2^22) and report the time-out.  dirty_out, when
// given, receives the number of reduction granules that are NOT the sentinel once everything enqueued on the context's stream has
// retired: 0 after any launch whose sweep completed, and 0 after a REPORTED time-out (the reporting path refills the buffer).
__global__ void kt_granules_dirty_kernel(const unsigned long long* __restrict__ g, int n, unsigned int* __restrict__ out)
{
    unsigned int dirty = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dirty += g[i] != KT_GRANULE_SENTINEL;
    if (dirty) atomicAdd(out, dirty);
}
extern "C" int kt_debug_handoff_fault(kt_ctx* c, int skip, int count, unsigned int spin_limit, unsigned int* dirty_out)
{
    KT_ARG(c && skip >= 0 && count >= 0);
    c->fault_skip = skip; c->fault_count = count;
    if (spin_limit) KT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(kt_sweep_spin_limit), &spin_limit, sizeof(spin_limit), 0, hipMemcpyHostToDevice, c->stream));
    if (dirty_out) {
        unsigned int* d = nullptr;
        KT_HIP(hipMalloc((void**)&d, sizeof(unsigned int)));
        KT_HIP(hipMemsetAsync(d, 0, sizeof(unsigned int), c->stream));
        // the two [15][256] kt_reduce29 sets (the second one starts at 32 * KT_RED_BLOCKS: kt_second_granules) and the level kernel's per-iteration sets
        for (int set = 0; set < 2; ++set)
            hipLaunchKernelGGL(kt_granules_dirty_kernel, dim3(4), dim3(256), 0, c->stream, (const unsigned long long*)c->red_partials + set * 32 * KT_RED_BLOCKS,
                               KT_RED_PAIRS * KT_RED_BLOCKS, d);
        for (int set = 0; set < KT_LEVEL_SETS; ++set)
            hipLaunchKernelGGL(kt_granules_dirty_kernel, dim3(4), dim3(256), 0, c->stream,
                               (const unsigned long long*)c->red_partials + KT_LEVEL_SET_BASE + set * KT_LEVEL_SET_STRIDE, KT_RED_PAIRS * KT_RED_BLOCKS, d);
        KT_HIP(hipMemcpyAsync(dirty_out, d, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        KT_HIP(hipStreamSynchronize(c->stream));
        KT_HIP(hipFree(d));
    }
    return KT_OK;
}


// which form the tracker's ICP-only odometry takes: one launch per iteration (0) or one per pyramid level (1); read when a tracker is created
#ifndef KT_ICP_LEVELS_DEFAULT
#define KT_ICP_LEVELS_DEFAULT 1
#endif
static int kt_icp_levels_override = -1;
extern "C" int kt_debug_icp_levels(int on) { kt_icp_levels_override = on < 0 ? -1 : (on != 0); return KT_OK; }
// The level kernel's workgroups wait for each other INSIDE the launch (a pose needs every workgroup's granules), so all KT_RED_GRID of them must
// be resident at once: a device (or a partition of one) that cannot hold the whole grid would run the first wave of workgroups into their bounded
// waits.  Checked once per process against the occupancy the runtime reports; such a device keeps the launch per iteration.
static bool kt_icp_levels_fit(int dev)
{
    static int fit[64];   // per device: 0 unknown, 1 fits, -1 does not (advisor, round 5: the answer was taken for the CURRENT device and kept for the process)
    if (dev < 0 || dev >= 64) return false;
    if (fit[dev] == 0) {
        int cus = 0, per_cu = 0, prev = 0;
        bool ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess;
        ok = ok && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
        ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kt_icp_level_kernel, KT_RED_THREADS, 0) == hipSuccess;
        (void)hipSetDevice(prev);
        fit[dev] = (ok && (long long)cus * per_cu >= KT_RED_GRID) ? 1 : -1;
    }
    return fit[dev] > 0;
}
bool kt_icp_levels_selected(int device)
{
    const char* e = getenv("KT_ICP_LEVELS");
    const bool env = e ? atoi(e) != 0 : KT_ICP_LEVELS_DEFAULT != 0;
    return (kt_icp_levels_override < 0 ? env : kt_icp_levels_override != 0) && kt_icp_levels_fit(device);
}
// -ri in the level form (kt_joint_level_kernel): off unless asked for (KT_RI_LEVELS=1 or the test hook)
static int kt_ri_levels_override = -1;
extern "C" int kt_debug_ri_levels(int on) { kt_ri_levels_override = on < 0 ? -1 : (on != 0); return KT_OK; }
bool kt_ri_levels_selected()
{
    const char* e = getenv("KT_RI_LEVELS");
    return kt_ri_levels_override < 0 ? (e && atoi(e) != 0) : kt_ri_levels_override != 0;
}
// the form was asked for explicitly (environment or test hook): policies that would pick one themselves keep out
bool kt_icp_levels_forced() { return kt_icp_levels_override >= 0 || getenv("KT_ICP_LEVELS") != nullptr; }
// test / tuning hook: the time bound of the hand-off waits in ticks of the 100 MHz clock (0 = the default, 50 ms)
extern "C" int kt_debug_wait_limit(kt_ctx* c, unsigned int ticks)
{
    KT_ARG(c);
    const unsigned int v = ticks ? ticks : 5000000u;
    KT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(kt_wait_limit_ticks), &v, sizeof(v), 0, hipMemcpyHostToDevice, c->stream));
    return KT_OK;
}
