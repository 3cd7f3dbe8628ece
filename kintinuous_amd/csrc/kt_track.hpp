// kt_track.hpp -- device-resident Gauss-Newton state and the per-iteration solve / pose update that the
// reference performs on the host (ICPOdometry.cpp:127-178, RGBDOdometry.cpp:213-231, 311-373,
// OdometryProvider.h:54-68), restated so it can run in the epilogue of the reduction kernels.
// The same inline functions are compiled for the host (frame set-up in kt_tracker.hip).
// Third-party math restated: Eigen LDLT<6x6 double> (pivoted, pseudo-inverse of D), cv::Rodrigues,
// Eigen Isometry3f compose / inverse (float, no FMA: the reference host build is -msse3).
#pragma once

#include "kt_common.hpp"

#include <float.h>
#include <math.h>

enum { KT_MODE_HOST = 0, KT_MODE_ICP_SOLVE = 1, KT_MODE_ICP_STASH = 2, KT_MODE_RGB_SOLVE = 3, KT_MODE_JOINT_SOLVE = 4 };

struct kt_level_k { double fx, fy, cx, cy; };  // IntrDoublePrecision at a pyramid level, internal.h:262-273

struct kt_track_state {
    float Rcurr[9], tcurr[3];                 // current estimate (ICPOdometry.cpp:73-74)
    float Rprev[9], tprev[3], Rprev_inv[9];   // previous pose and its inverse (:70-83)
    double resultRt[16];                      // accumulated increment, carried across levels (:85, :144)
    float krkinv[9], kt[3];                   // K R K^-1 and K t for computeRgbResidual (RGBDOdometry.cpp:213-231)
    float sigma_val;                          // RGBDOdometry.cpp:253
    int rgb_count, rgb_sigma;
    float icp29[29];                          // ICP sums stashed for the joint solve
    float last_residual[2];
    int handoff_timeout;                      // set by a reduction epilogue whose hand-off sweep gave up
    int fusion_skipped;                       // kt_frame_setup_kernel: the frame needs a volume shift, fusion kernels did nothing
    int pad2;
};

#ifdef __HIPCC__
#define KT_HD __host__ __device__ __forceinline__
#else
#define KT_HD inline
#endif

// reduce.cu:401-418 unpack, straight into the double system of ICPOdometry.cpp:127-128
KT_HD void kt_unpack29_d(const float* h, double* A, double* b)
{
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const double value = (double)h[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
}

// Eigen LDLT<Matrix<double,6,6>>::compute + solve: pivot on the largest remaining diagonal, D^-1 as a
// pseudo-inverse (entries below max|D| * eps are dropped).  A is overwritten.
KT_HD void kt_ldlt_solve6(double* A, const double* bin, double* x)
{
    const int n = 6;
    int tr[6];
    for (int k = 0; k < n; ++k) {
        int p = k;
        double big = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i * n + i]) > big) { big = fabs(A[i * n + i]); p = i; }
        tr[k] = p;
        if (p != k) {
            for (int j = 0; j < n; ++j) { const double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            for (int i = 0; i < n; ++i) { const double t = A[i * n + k]; A[i * n + k] = A[i * n + p]; A[i * n + p] = t; }
        }
        double d = A[k * n + k];
        for (int j = 0; j < k; ++j) d -= A[k * n + j] * A[k * n + j] * A[j * n + j];
        A[k * n + k] = d;
        for (int i = k + 1; i < n; ++i) {
            double s = A[i * n + k];
            for (int j = 0; j < k; ++j) s -= A[i * n + j] * A[k * n + j] * A[j * n + j];
            A[i * n + k] = (d != 0.0) ? s / d : s;
        }
    }
    for (int i = 0; i < n; ++i) x[i] = bin[i];
    for (int k = 0; k < n; ++k)
        if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) x[i] -= A[i * n + j] * x[j];
    double maxd = 0;
    for (int i = 0; i < n; ++i)
        if (fabs(A[i * n + i]) > maxd) maxd = fabs(A[i * n + i]);
    double tol = maxd * DBL_EPSILON;
    if (tol < 1.0 / DBL_MAX) tol = 1.0 / DBL_MAX;
    for (int i = 0; i < n; ++i) x[i] = (fabs(A[i * n + i]) > tol) ? x[i] / A[i * n + i] : 0.0;
    for (int i = n - 1; i >= 0; --i)
        for (int j = i + 1; j < n; ++j) x[i] -= A[j * n + i] * x[j];
    for (int k = n - 1; k >= 0; --k)
        if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
}

// cv::Rodrigues (rotation vector -> matrix), OdometryProvider.h:54-68
KT_HD void kt_rodrigues(const double* r, double* R)
{
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
}

// Eigen Matrix3f::inverse() (cofactor form), ICPOdometry.cpp:81
KT_HD void kt_mat33_inverse(const float* m, float* out)
{
#define KT_M(i, j) m[(i) * 3 + (j)]
#define KT_COF(i, j) (KT_M(((i) + 1) % 3, ((j) + 1) % 3) * KT_M(((i) + 2) % 3, ((j) + 2) % 3) - KT_M(((i) + 1) % 3, ((j) + 2) % 3) * KT_M(((i) + 2) % 3, ((j) + 1) % 3))
    const float c00 = KT_COF(0, 0), c10 = KT_COF(1, 0), c20 = KT_COF(2, 0);
    const float det = (c00 * KT_M(0, 0) + c10 * KT_M(1, 0)) + c20 * KT_M(2, 0);
    const float invdet = 1.0f / det;
    float r[9];
    r[0] = c00 * invdet; r[1] = c10 * invdet; r[2] = c20 * invdet;
    r[3] = KT_COF(0, 1) * invdet; r[4] = KT_COF(1, 1) * invdet; r[5] = KT_COF(2, 1) * invdet;
    r[6] = KT_COF(0, 2) * invdet; r[7] = KT_COF(1, 2) * invdet; r[8] = KT_COF(2, 2) * invdet;
#undef KT_COF
#undef KT_M
    for (int k = 0; k < 9; ++k) out[k] = r[k];
}

// x (6, double) -> resultRt = [Rodrigues | t] * resultRt;  T_curr = T_prev * inverse([R | t]) in float
// (ICPOdometry.cpp:133-178 == RGBDOdometry.cpp:328-373)
KT_HD void kt_pose_update(const double* x, double* resultRt, const float* Rprev, const float* tprev, float* Rcurr, float* tcurr)
{
    double currRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double R[9];
    kt_rodrigues(&x[3], R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) currRt[i * 4 + j] = R[i * 3 + j];
        currRt[i * 4 + 3] = x[i];
    }
    double prod[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += currRt[i * 4 + k] * resultRt[k * 4 + j];
            prod[i * 4 + j] = s;
        }
    for (int k = 0; k < 16; ++k) resultRt[k] = prod[k];
    float rot[9], trans[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) rot[i * 3 + j] = (float)resultRt[i * 4 + j];
        trans[i] = (float)resultRt[i * 4 + 3];
    }
    float rinv[9], tinv[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rinv[i * 3 + j] = rot[j * 3 + i];
    for (int i = 0; i < 3; ++i) tinv[i] = -((rinv[i * 3 + 0] * trans[0] + rinv[i * 3 + 1] * trans[1]) + rinv[i * 3 + 2] * trans[2]);
    float Rn[9], tn[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = (Rprev[i * 3 + 0] * rinv[0 * 3 + j] + Rprev[i * 3 + 1] * rinv[1 * 3 + j]) + Rprev[i * 3 + 2] * rinv[2 * 3 + j];
    for (int i = 0; i < 3; ++i) tn[i] = ((Rprev[i * 3 + 0] * tinv[0] + Rprev[i * 3 + 1] * tinv[1]) + Rprev[i * 3 + 2] * tinv[2]) + tprev[i];
    for (int k = 0; k < 9; ++k) Rcurr[k] = Rn[k];
    for (int k = 0; k < 3; ++k) tcurr[k] = tn[k];
}

// K R K^-1 and K t from the inverse of the accumulated increment (RGBDOdometry.cpp:213-231)
KT_HD void kt_compute_krk(const double* resultRt, const kt_level_k k, float* krkinv, float* kt)
{
    const double K[9] = {k.fx, 0, k.cx, 0, k.fy, k.cy, 0, 0, 1};
    const double Kinv[9] = {1.0 / K[0], 0, -K[2] / K[0], 0, 1.0 / K[4], -K[5] / K[4], 0, 0, 1};
    double Rinv[9], tinv[3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rinv[a * 3 + b] = resultRt[b * 4 + a];
    for (int a = 0; a < 3; ++a) tinv[a] = -(Rinv[a * 3 + 0] * resultRt[3] + Rinv[a * 3 + 1] * resultRt[7] + Rinv[a * 3 + 2] * resultRt[11]);
    double KR[9], KRK[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int q = 0; q < 3; ++q) s += K[i * 3 + q] * Rinv[q * 3 + j];
            KR[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int q = 0; q < 3; ++q) s += KR[i * 3 + q] * Kinv[q * 3 + j];
            KRK[i * 3 + j] = s;
        }
    for (int a = 0; a < 3; ++a) kt[a] = (float)(K[a * 3 + 0] * tinv[0] + K[a * 3 + 1] * tinv[1] + K[a * 3 + 2] * tinv[2]);
    for (int n = 0; n < 9; ++n) krkinv[n] = (float)KRK[n];
}

#ifdef __HIPCC__
// Device-side 6x6 solve for the kernel epilogues.  (A register-resident version with in-loop pivoting (if-converted swaps, ~1300
// instructions) and a one-row-per-lane version with v_readlane / ds_bpermute exchange were both measured at 2.4-2.6 us; the hoisted
// form below takes 2.2 us and is the simplest of the three.)
// Pivot-hoisted variant.  Eigen's unblocked LDLT is left-looking: step k only touches column k, so the diagonal entries it pivots on
// at step k (rows >= k) are still the ORIGINAL ones -- the whole pivot sequence is a selection sort of the original diagonal (first
// maximum wins, swaps included) and can be decided before any elimination arithmetic.  So: (1) simulate the swaps on the 6
// diagonal values, (2) gather the symmetrically permuted system from LDS with dynamic addresses, (3) factorise and solve
// with compile-time indices and no swaps at all, (4) scatter x back through the permutation.  Every element sees exactly the
// operations of kt_ldlt_solve6 in the same order; the code is ~40% shorter than with in-loop swaps.
// The system arrives IN LDS (round 3): 42 lanes of the sweeping workgroup each convert / combine one element (kt_sys_slot), so the
// solving thread never holds the unpermuted matrix in registers -- with it the RGB-D epilogues (two unpacked systems live at once)
// spilled 66-84 VGPRs to scratch in the serial tail.  sys: [0, 36) A row-major, [36, 42) b, [42, 48) scratch of the calling thread.
#define KT_SYS_DOUBLES 48
// packed index (reduce.cu:401-418: rows i = 0..5, columns j = i..6 in order) of element k of the LDS system
__device__ __forceinline__ int kt_sys_slot(int k)
{
    int i = k / 6, j = k - 6 * i;
    if (k >= 36) { i = k - 36; j = 6; }
    else if (i > j) { const int t = i; i = j; j = t; }
    return i * 7 - (i * (i - 1)) / 2 + (j - i);
}
__device__ __forceinline__ void kt_ldlt_solve6_hoisted(double* sys, double (&x)[6])
{
    // (1) pivot order
    double dg[6];
    int idx[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { dg[i] = sys[i * 7]; idx[i] = i; }
#pragma unroll
    for (int K = 0; K < 5; ++K) {
        int p = K;
        double big = fabs(dg[K]);
#pragma unroll
        for (int i = K + 1; i < 6; ++i) {
            const double v = fabs(dg[i]);
            if (v > big) { big = v; p = i; }
        }
        p = __builtin_amdgcn_readfirstlane(p);
#pragma unroll
        for (int c = K + 1; c < 6; ++c)
            if (p == c) {
                asm volatile("; pivot swap" ::: "memory");
                const double t = dg[K]; dg[K] = dg[c]; dg[c] = t;
                const int ti = idx[K]; idx[K] = idx[c]; idx[c] = ti;
            }
    }
    // (2) permuted system A' = P A P^T (lower triangle), b' = P b
    double A[36];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) A[i * 6 + j] = sys[idx[i] * 6 + idx[j]];
        x[i] = sys[36 + idx[i]];
    }
    // (3) LDL^T without pivoting, forward / diagonal / backward substitution
#pragma unroll
    for (int K = 0; K < 6; ++K) {
        double d = A[K * 6 + K];
#pragma unroll
        for (int j = 0; j < K; ++j) d -= A[K * 6 + j] * A[K * 6 + j] * A[j * 6 + j];
        A[K * 6 + K] = d;
#pragma unroll
        for (int i = K + 1; i < 6; ++i) {
            double sacc = A[i * 6 + K];
#pragma unroll
            for (int j = 0; j < K; ++j) sacc -= A[i * 6 + j] * A[K * 6 + j] * A[j * 6 + j];
            A[i * 6 + K] = (d != 0.0) ? sacc / d : sacc;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) x[i] -= A[i * 6 + j] * x[j];
    double maxd = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (fabs(A[i * 6 + i]) > maxd) maxd = fabs(A[i * 6 + i]);
    double tol = maxd * DBL_EPSILON;
    if (tol < 1.0 / DBL_MAX) tol = 1.0 / DBL_MAX;
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = (fabs(A[i * 6 + i]) > tol) ? x[i] / A[i * 6 + i] : 0.0;
#pragma unroll
    for (int i = 5; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < 6; ++j) x[i] -= A[j * 6 + i] * x[j];
    // (4) x = P^T x'
#pragma unroll
    for (int i = 0; i < 6; ++i) sys[42 + idx[i]] = x[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = sys[42 + i];
}

// opt-in timing probes of the reduction tail (build with KT_EXTRA_FLAGS=-DKT_ICP_TIMING; scripts/icp_timing.py)
#ifdef KT_ICP_TIMING
#define KT_TS(i) do { if (threadIdx.x == 0) { kt_ts[i] = wall_clock64(); if ((i) <= 2) kt_wg_ts[(i) * 256 + (blockIdx.x & 255)] = kt_ts[i]; } } while (0)
__shared__ unsigned long long kt_ts[8];
__device__ unsigned long long kt_wg_ts[3 * 256];   // per workgroup of the last reduction launch: loop entered, loop done, granules published (100 MHz ticks)
#else
#define KT_TS(i) do {} while (0)
#endif

// the part of the device state one Gauss-Newton step reads, held in registers by the solving thread
struct kt_pose_regs {
    double resultRt[16];
    float Rprev[9], tprev[3];
    __device__ __forceinline__ void load(const double* d, const float* f)
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) resultRt[k] = d[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rprev[k] = f[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tprev[k] = f[9 + k];
    }
};
// ... fetched EARLY by the sweeping workgroup (before the sweep, so that the latency hides under it), one element per lane of the first
// 28 -- the solving thread alone would hold 44 VGPRs across the sweep -- and handed over through LDS with the 6x6 system.
#define KT_POSE_STAGE_DOUBLES 16
#define KT_POSE_STAGE_FLOATS 12
#define KT_TAIL_WORK_DOUBLES 36   // LDS scratch of kt_solve_and_update_wave: the 6 x 6 factor (its transpose is read back), then the float increment
struct kt_pose_stage {
    double d;
    float f;
    __device__ __forceinline__ void fetch(const kt_track_state* st)
    {
        const int t = threadIdx.x;
        d = 0.0; f = 0.0f;
        if (t < 16) d = st->resultRt[t];
        else if (t < 25) f = st->Rprev[t - 16];
        else if (t < 28) f = st->tprev[t - 25];
    }
    __device__ __forceinline__ void park(double* lds_d, float* lds_f) const   // followed by a __syncthreads() of the caller
    {
        const int t = threadIdx.x;
        if (t < 16) lds_d[t] = d;
        else if (t < 28) lds_f[t - 16] = f;
    }
};

// executed by ONE thread in the epilogue of the sweeping workgroup
__device__ __forceinline__ void kt_solve_and_update(kt_track_state* st, kt_pose_regs& pr, double* sys)
{
    double x[6];
    kt_ldlt_solve6_hoisted(sys, x);
    KT_TS(5);
    float Rcurr[9], tcurr[3];
    kt_pose_update(x, pr.resultRt, pr.Rprev, pr.tprev, Rcurr, tcurr);
    KT_TS(6);
#pragma unroll
    for (int k = 0; k < 16; ++k) st->resultRt[k] = pr.resultRt[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) st->Rcurr[k] = Rcurr[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) st->tcurr[k] = tcurr[k];
}
__device__ __forceinline__ void kt_update_krk(kt_track_state* st, const kt_level_k k) { kt_compute_krk(st->resultRt, k, st->krkinv, st->kt); }

// ------------------------------------------------------------------------------------------------
// The same tail spread over the LANES of one wave (round 4).  The serial form above is ~1300 executed instructions of one thread, and a
// lone wave issues a dependent f64 operation only every 6-8 cycles: 3.7 us per Gauss-Newton iteration whatever the rest of the chip does
// (profiles/r04_experiments.md: not instruction fetch -- a second pass over the same code takes 3.4 us).  Its cost is its instruction
// count, so the arithmetic goes sideways:
//   * LDL^T: lane r owns row r of the permuted system.  Eigen's unblocked algorithm is left-looking, so at step K every lane i >= K forms
//     A[i][K] - sum_j (A[i][j] A[K][j]) D[j] for its own row -- lane K's result is the pivot d_K -- with row K and the pivots broadcast by
//     v_readlane, and ONE division per step serves the five rows below (6 divisions instead of 21, 45 multiply-subtract terms instead of 105);
//   * forward substitution by lanes (x[i] -= L[i][j] x[j], j ascending: x[j] is final when it is needed); backward substitution is
//     serial by construction (x[i] subtracts L[j][i] x[j] for j ASCENDING, and its first operand is the one that is ready last): the
//     products are formed once per column, the subtractions by the lane that owns x[i];
//   * Rodrigues row i, the row-i-times-column-j products of the 4 x 4 composition and the nine + three elements of the float pose each on
//     their own lane; the state leaves in three store instructions.
// Every element sees exactly the operations of kt_ldlt_solve6 / kt_pose_update in the same order (the bit-exact trajectory tests and
// kt_debug_solve_check -- both forms on the same systems, ties, zeros, NaNs -- compare them).
// Called by all 64 lanes of ONE wave with wave-uniform arguments.  sys: [0, 36) A row-major, [36, 42) b, [42, 48) scratch; pose_d:
// resultRt[16]; pose_f: Rprev[9], tprev[3]; work: KT_TAIL_WORK_DOUBLES (36) doubles of LDS scratch.  On return pose_d holds the new resultRt (for kt_compute_krk).
// ------------------------------------------------------------------------------------------------
#ifdef KT_TAIL_MARK   // analysis builds: section markers in the ISA (scripts/isa_summary.py counts the instructions between them)
#define KT_MARK(n) asm volatile("; TAILSEC " #n ::: "memory")
#else
#define KT_MARK(n) do {} while (0)
#endif
__device__ __forceinline__ double kt_lane_bcast(double v, int src)   // src: wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// gran != nullptr (kt_icp_level_kernel: several iterations in one launch): the new pose also leaves as 12 tagged granules {float, seq} for the
// other workgroups, which poll them ("the data is the flag", as in kt_reduce29); pose_lds != nullptr: and as 12 floats in LDS for this workgroup
// itself (kt_joint_level_kernel passes pose_lds alone and publishes pose AND K R K^-1 / K t together, behind its own barrier).
__device__ __forceinline__ void kt_solve_and_update_wave(kt_track_state* st, double* sys, double* pose_d, const float* pose_f, double* work,
                                                         unsigned long long* gran = nullptr, unsigned int seq = 0, float* pose_lds = nullptr)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int r = min(lane, 5);   // lanes past the sixth repeat row 5 (their results are never used)
    KT_MARK(1);
    // (1) pivot order.  Lane l keeps ORIGINAL row l; what is permuted is its logical position pos (the row of A' = P A P^T it stands
    // for) and the order idx[] in which everybody walks the columns.  The order is the selection sort of the original diagonal (see
    // kt_ldlt_solve6_hoisted).  Without ties it is simply the rank of |A[l][l]| -- counted by the lanes, no swaps; with a tie or a NaN on
    // the diagonal the swap history decides, and the serial emulation runs (wave-uniform branch).
    double dg[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dg[i] = fabs(sys[i * 7]);
    const double mydg = fabs(sys[r * 7]);
    int n_gt = 0, n_ge = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { n_gt += (dg[i] > mydg) ? 1 : 0; n_ge += (dg[i] >= mydg) ? 1 : 0; }
    int pos = n_gt;
    int idx[6];
    if (__builtin_amdgcn_ballot_w64(n_ge != n_gt + 1) == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) idx[c] = __builtin_ctzll(__builtin_amdgcn_ballot_w64((n_gt == c) & (lane < 6)) | (1ull << 63));
    } else {
        asm volatile("; pivot order with ties" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) idx[i] = i;
#pragma unroll
        for (int K = 0; K < 5; ++K) {
            int p = K;
            double big = dg[K];
#pragma unroll
            for (int i = K + 1; i < 6; ++i)
                if (dg[i] > big) { big = dg[i]; p = i; }
            p = __builtin_amdgcn_readfirstlane(p);
#pragma unroll
            for (int c = K + 1; c < 6; ++c)
                if (p == c) {
                    asm volatile("; pivot swap" ::: "memory");
                    const double t = dg[K]; dg[K] = dg[c]; dg[c] = t;
                    const int ti = idx[K]; idx[K] = idx[c]; idx[c] = ti;
                }
        }
        pos = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { idx[i] = __builtin_amdgcn_readfirstlane(idx[i]); pos = (idx[i] == r) ? i : pos; }
    }
    KT_MARK(2);
    // (2) row l with its columns in pivot order, and b[l]
    double a[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) a[c] = sys[r * 6 + idx[c]];
    double x = sys[36 + r];
    KT_MARK(3);
    // (3) LDL^T, left-looking; D[] and the broadcast row are wave-uniform.  Step K: the lane at position K owns the pivot.
    double D[6];
#pragma unroll
    for (int K = 0; K < 6; ++K) {
        double s = a[K];
#pragma unroll
        for (int j = 0; j < K; ++j) s -= a[j] * kt_lane_bcast(a[j], idx[K]) * D[j];
        const double d = kt_lane_bcast(s, idx[K]);
        D[K] = d;
        if (K < 5) {
            const double q = (d != 0.0) ? s / d : s;
            a[K] = (pos > K) ? q : s;
        } else {
            a[K] = s;
        }
    }
    KT_MARK(31);
    // the factor by positions, for the backward substitution (its columns) and the lane's own pivot: T[pos][c]
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) {
#pragma unroll
        for (int c = 0; c < 6; ++c) work[pos * 6 + c] = a[c];
    }
    __builtin_amdgcn_wave_barrier();
    KT_MARK(4);
    // (4) forward substitution
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const double xj = kt_lane_bcast(x, idx[j]);
        const double y = x - a[j] * xj;
        x = (pos > j) ? y : x;
    }
    KT_MARK(5);
    // (5) D^-1 as a pseudo-inverse
    double maxd = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (fabs(D[i]) > maxd) maxd = fabs(D[i]);
    double tol = maxd * DBL_EPSILON;
    if (tol < 1.0 / DBL_MAX) tol = 1.0 / DBL_MAX;
    const double mine = work[pos * 7];
    x = (fabs(mine) > tol) ? x / mine : 0.0;
    KT_MARK(6);
    // (6) backward substitution: x[i] -= L[j][i] x[j], j ascending
    double cl[6];
#pragma unroll
    for (int j = 1; j < 6; ++j) cl[j] = work[j * 6 + pos];
    double X[6], P[6];
    X[5] = kt_lane_bcast(x, idx[5]);
#pragma unroll
    for (int s = 4; s >= 0; --s) {
        P[s + 1] = cl[s + 1] * X[s + 1];
        double y = x;
#pragma unroll
        for (int j = s + 1; j < 6; ++j) y -= P[j];
        x = (pos == s) ? y : x;
        if (s > 0) X[s] = kt_lane_bcast(x, idx[s]);
    }
    KT_MARK(7);
    // (7) lane l holds the unknown l: no permutation to undo
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) sys[42 + lane] = x;
    __builtin_amdgcn_wave_barrier();
    double xs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xs[i] = sys[42 + i];
    KT_TS(5);
    KT_MARK(8);
    // (8) cv::Rodrigues (kt_rodrigues), row i of [R | t] on the lanes of row i; i == 3: (0, 0, 0, 1)
    const int pi = (lane >> 2) & 3, pj = lane & 3;
    double crow[4];
    {
        const double theta = sqrt(xs[3] * xs[3] + xs[4] * xs[4] + xs[5] * xs[5]);
        if (theta < DBL_EPSILON) {
#pragma unroll
            for (int k = 0; k < 3; ++k) crow[k] = (pi == k) ? 1.0 : 0.0;
        } else {
            const double c = cos(theta), sn = sin(theta), c1 = 1. - c, itheta = 1. / theta;
            const double rx = xs[3] * itheta, ry = xs[4] * itheta, rz = xs[5] * itheta;
            const double rv[3] = {rx, ry, rz};
            const double ri = (pi == 0) ? rx : ((pi == 1) ? ry : rz);
            const double cI1 = c * 1.0, cI0 = c * 0.0;
            // r_x = {0, -rz, ry;  rz, 0, -rx;  -ry, rx, 0}
            const double rxk[3] = {(pi == 0) ? 0.0 : ((pi == 1) ? rz : -ry), (pi == 0) ? -rz : ((pi == 1) ? 0.0 : rx), (pi == 0) ? ry : ((pi == 1) ? -rx : 0.0)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double rrt = ri * rv[k];   // (kt_rodrigues forms r_min * r_max: the same product)
                crow[k] = ((pi == k) ? cI1 : cI0) + c1 * rrt + sn * rxk[k];
            }
        }
        crow[3] = (pi == 0) ? xs[0] : ((pi == 1) ? xs[1] : xs[2]);
        if (pi == 3) { crow[0] = 0.0; crow[1] = 0.0; crow[2] = 0.0; crow[3] = 1.0; }
    }
    KT_MARK(9);
    // (the addresses of the tail's stores are formed HERE, from the scalar base: hoisted to the kernel's prologue they were lane-dependent
    // 64-bit values alive across the whole pixel loop, and the register allocator of the multi-level kernel spilled them to scratch -- reloaded
    // in front of every store, on the critical path)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));   // (the stores below index with lane_t: their addresses cannot be formed before this point)
    // (9) resultRt = [R | t] * resultRt: element (pi, pj) on lane 4 pi + pj
    double prod = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) prod += crow[k] * pose_d[k * 4 + pj];
    __builtin_amdgcn_wave_barrier();
    float* fw = (float*)work;   // [0, 16): the new increment in float
    if (lane < 16) {
        st->resultRt[lane_t] = prod;
        pose_d[lane] = prod;
        fw[lane] = (float)prod;
    }
    __builtin_amdgcn_wave_barrier();
    KT_MARK(10);
    // (10) T_curr = T_prev * inverse([R | t]) in float (kt_pose_update): lanes 0..8 the rotation, 9..11 the translation
    float rot[9], trans[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) rot[i * 3 + j] = fw[i * 4 + j];
        trans[i] = fw[i * 4 + 3];
    }
    float tinv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tinv[i] = -((rot[0 * 3 + i] * trans[0] + rot[1 * 3 + i] * trans[1]) + rot[2 * 3 + i] * trans[2]);
    const bool is_t = lane >= 9;
    const int l12 = min(lane, 11);
    const int oi = is_t ? l12 - 9 : (l12 * 11) >> 5, oj = l12 - 3 * oi;   // (lanes 9..11: oj unused)
    float B[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float rjk = (oj == 0) ? rot[0 * 3 + k] : ((oj == 1) ? rot[1 * 3 + k] : rot[2 * 3 + k]);   // rinv[k][j] = rot[j][k]
        B[k] = is_t ? tinv[k] : rjk;
    }
    const float p0 = pose_f[oi * 3 + 0], p1 = pose_f[oi * 3 + 1], p2 = pose_f[oi * 3 + 2];
    float v = (p0 * B[0] + p1 * B[1]) + p2 * B[2];
    const float vt = v + pose_f[9 + oi];
    v = is_t ? vt : v;
    if (lane < 12) (&st->Rcurr[0])[lane_t] = v;   // Rcurr[9] and tcurr[3] are adjacent in kt_track_state
    if (pose_lds && lane < 12) pose_lds[lane_t] = v;   // the resident kernels' own copy (kt_icp_level_kernel, kt_joint_level_kernel)
    if (gran) {
        // (kt_icp_level_kernel: the other waves of the workgroup arrive here once their hand-back stores have completed -- the pose must not be
        // observable before the granules it will be answered into are sentinels again; this wave sweeps nothing and has none of its own)
        __builtin_amdgcn_s_barrier();
        if (lane < 12)
            __hip_atomic_store(&gran[lane_t], ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    KT_MARK(11);
    KT_TS(6);
}
static_assert(offsetof(kt_track_state, tcurr) == offsetof(kt_track_state, Rcurr) + 9 * sizeof(float), "Rcurr / tcurr adjacency");
#endif
