// kt_slice.hip -- the per-slice stage of the reference's CloudSliceProcessor (backend/CloudSliceProcessor.cpp:87-163) on the GPU:
// weight cull (alpha >= weightCull), pcl::VoxelGrid<PointXYZRGB> down-sampling at the voxel leaf size, pcl::NormalEstimation with the 20
// nearest neighbours, output pcl::PointXYZRGBNormal.  It is the step right behind every volume shift and the first place the reference's
// backend falls behind ("map lagging behind", README.md:184-186).
// PCL 1.7 is not vendored with the reference: the arithmetic restates its published algorithms (filters/impl/voxel_grid.hpp applyFilter,
// common/impl/centroid.hpp computeMeanAndCovarianceMatrix in its float single-pass form, features/normal_3d.h solvePlaneParameters and
// flipNormalTowardsViewpoint with the default sensor origin, common/impl/eigen.hpp computeRoots / eigen33), float, no contraction.
// Two orders PCL leaves to its implementation are fixed (as in oracle/kt_oracle_kernels.c): the points of a leaf are summed in their
// original order (a stable sort instead of std::sort), neighbours are ordered by (squared distance, index).
//
// Shape on the GPU -- a device-resident stage (kt_slice_process_device): the points stay where the extraction kernel left them, every
// intermediate (the number of points, the bounding box, the leaf grid, the number of leaves) lives in device memory and is read by the
// next kernel from there, so the whole stage is ONE uninterrupted run of launches on the workspace's own stream; the host learns the
// output count from a pinned word behind the last launch.  Steps: bounding box of the points that pass the weight cull; leaf keys (a
// culled point gets the key 0xffffffff and sorts to the end: the cull needs no compaction pass); ONE stable radix sort of (key, index)
// pairs and one scan of the run heads (rocPRIM's device-wide primitives, called directly: library plumbing, everything arithmetic is
// written here); one thread per leaf walks its short run in order; after the grid every point owns a distinct leaf cell, so the k
// nearest neighbours of a point are found by binary-searching the sorted leaf keys of the (2r + 1)^3 cells around its own cell -- all
// points within r leaf sizes are in there -- with r growing until the k-th neighbour is provably the k-th nearest (distance <= r *
// leaf), then covariance, smallest eigenvector and curvature in the same thread.  The workspace (kt_slice_ws) is allocated once for a
// capacity and reused: no allocation, no copy-back and no synchronisation inside a call.
#include "kt_internal.hpp"

#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#define KT_SLICE_K_MAX 64

namespace {

struct BBox { float mn[3], mx[3]; };
struct Grid { int min_b[3], div_b[3]; float inv_leaf, leaf; };
// what the kernels of one call hand to each other, in device memory
struct Params { Grid g; int gridded; unsigned int leaves; };
#define KT_SLICE_CULLED 0xffffffffu   // key of a point that fails the weight cull (or lies past the input's end): sorts behind every leaf
#define KT_SLICE_BOXES 256

__device__ __forceinline__ bool slice_kept(const kt_point_xyzrgb& p, int weight_cull) { return !(weight_cull > 0) || (int)p.a >= weight_cull; }

// getMinMax3D over the points that pass the cull (CloudSliceProcessor.cpp:99-117 runs the cull first): per-workgroup partial boxes
__global__ __launch_bounds__(256) void slice_bbox(const kt_point_xyzrgb* __restrict__ pts, const unsigned int* __restrict__ n_dev, int n_max, int weight_cull,
                                                  BBox* __restrict__ partial)
{
    __shared__ BBox sh[4];
    const int n = (int)min(*n_dev, (unsigned int)n_max);   // (an extraction counts past its buffer's capacity)
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const kt_point_xyzrgb p = pts[i];
        if (!slice_kept(p, weight_cull)) continue;
        const float v[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], v[a]); mx[a] = fmaxf(mx[a], v[a]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64)); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) { sh[threadIdx.x >> 6].mn[a] = mn[a]; sh[threadIdx.x >> 6].mx[a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        BBox b = sh[0];
        for (int w = 1; w < 4; ++w)
            for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], sh[w].mn[a]); b.mx[a] = fmaxf(b.mx[a], sh[w].mx[a]); }
        partial[blockIdx.x] = b;
    }
}

// one wave: fold the partial boxes, derive the leaf grid (voxel_grid.hpp applyFilter: min_b / max_b / div_b, the int32 overflow check)
__global__ __launch_bounds__(64) void slice_grid(const BBox* __restrict__ partial, float leaf, Params* __restrict__ prm)
{
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = threadIdx.x; i < KT_SLICE_BOXES; i += 64)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], partial[i].mn[a]); mx[a] = fmaxf(mx[a], partial[i].mx[a]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64)); }
    if (threadIdx.x != 0) return;
    Params p;
    p.g.leaf = leaf;
    p.g.inv_leaf = 1.0f / leaf;
    p.leaves = 0;
    long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        p.g.min_b[a] = (int)floorf(mn[a] * p.g.inv_leaf);
        p.g.div_b[a] = (int)floorf(mx[a] * p.g.inv_leaf) - p.g.min_b[a] + 1;
        // voxel_grid.hpp: dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inverse_leaf_size_[0]) + 1, ...; dx * dy * dz > INT32_MAX
        cells *= (long long)((mx[a] - mn[a]) * p.g.inv_leaf) + 1;
    }
    // "Leaf size is too small for the input dataset. Integer indices would overflow.": PCL passes the cloud through unfiltered
    p.gridded = !(mn[0] <= mx[0]) || cells <= 2147483647LL;   // (no point kept: nothing to pass through either)
    *prm = p;
}

__global__ __launch_bounds__(256) void slice_keys(const kt_point_xyzrgb* __restrict__ pts, const unsigned int* __restrict__ n_dev, int n_max, int weight_cull,
                                                  const Params* __restrict__ prm, unsigned int* __restrict__ keys, unsigned int* __restrict__ src)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_max) return;
    src[i] = (unsigned int)i;
    if ((unsigned int)i >= *n_dev) { keys[i] = KT_SLICE_CULLED; return; }
    const kt_point_xyzrgb p = pts[i];
    if (!slice_kept(p, weight_cull)) { keys[i] = KT_SLICE_CULLED; return; }
    const Grid g = prm->g;
    if (!prm->gridded) { keys[i] = (unsigned int)i; return; }   // every point its own "leaf", in input order
    // voxel_grid.hpp: ijk = static_cast<int>(floor(p * inverse_leaf_size) - static_cast<float>(min_b)); idx = ijk . divb_mul
    const int i0 = (int)(__builtin_floorf(p.x * g.inv_leaf) - (float)g.min_b[0]);
    const int i1 = (int)(__builtin_floorf(p.y * g.inv_leaf) - (float)g.min_b[1]);
    const int i2 = (int)(__builtin_floorf(p.z * g.inv_leaf) - (float)g.min_b[2]);
    keys[i] = (unsigned int)(i0 + i1 * g.div_b[0] + i2 * g.div_b[0] * g.div_b[1]);
}

__global__ __launch_bounds__(256) void slice_heads(const unsigned int* __restrict__ keys, int n_max, unsigned int* __restrict__ head)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_max) head[i] = (keys[i] != KT_SLICE_CULLED && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}

// one thread per leaf (= per run of equal keys in the sorted order): the centroid of x, y, z and of r, g, b as floats, summed in run order
__global__ __launch_bounds__(256) void slice_centroids(const kt_point_xyzrgb* __restrict__ pts, const unsigned int* __restrict__ keys,
                                                       const unsigned int* __restrict__ src, const unsigned int* __restrict__ head,
                                                       const unsigned int* __restrict__ leaf_of, int n_max, float* __restrict__ cen,
                                                       unsigned int* __restrict__ leaf_key, unsigned int* __restrict__ leaf_src, Params* __restrict__ prm)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_max) return;
    if (i == n_max - 1) prm->leaves = leaf_of[i];   // inclusive scan of the head flags: its last entry is the number of leaves
    if (!head[i]) return;
    const unsigned int key = keys[i];
    float acc[6] = {0, 0, 0, 0, 0, 0};
    int j = i;
    for (; j < n_max && keys[j] == key; ++j) {
        const kt_point_xyzrgb p = pts[src[j]];
        acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += (float)p.r; acc[4] += (float)p.g; acc[5] += (float)p.b;
    }
    // `centroid /= static_cast<float>(count)` on an Eigen::VectorXf: Eigen 3.2 (the reference's, README.md:14-31) evaluates a
    // floating-point `/= s` as a multiplication by Scalar(1) / s (SelfCwiseBinaryOp.h; true division only from 3.3 on)
    const float inv_cnt = 1.0f / (float)(j - i);
    const unsigned int q = leaf_of[i] - 1u;
#pragma unroll
    for (int a = 0; a < 6; ++a) cen[(size_t)q * 6 + a] = acc[a] * inv_cnt;
    leaf_key[q] = key;
    leaf_src[q] = src[i];   // the leaf's first point (the pass-through case copies its weight byte)
}

// ---- pcl::eigen33 / computeRoots (common/impl/eigen.hpp), float ----
// sin / cos / atan2 of pcl::computeRoots, on the small domain the stage needs (atan2(y >= 0, x) in [0, pi], sin / cos on [0, pi / 3]):
// the Cephes single-precision kernels, the SAME operations in the SAME order as oracle/kt_oracle_kernels.c (sp_atan01, sp_atan2_pos,
// sp_sincos), every fused multiply-add written out -- the device library's atan2f / sinf / cosf differ from libm in the last bits,
// which made the normals the one output of the stage that was only close to the oracle's (1e-4) instead of equal.
__device__ __forceinline__ float sp_atan01(float a)
{
    float y0 = 0.0f, x = a;
    if (a > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (a - 1.0f) / (a + 1.0f); }
    const float z = x * x;
    float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y0 + __builtin_fmaf(p * z, x, x);
}
__device__ __forceinline__ float sp_atan2_pos(float y, float x)
{
    const float ax = fabsf(x);
    const float hi = ax > y ? ax : y, lo = ax > y ? y : ax;
    if (!(hi > 0.0f)) return 0.0f;
    float r = sp_atan01(lo / hi);
    if (y > ax) r = 1.57079632679489661923f - r;
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return r;
}
__device__ __forceinline__ void sp_sincos(float t, float& s, float& c)
{
    const bool swap = t > 0.78539816339744830962f;
    const float x = swap ? 1.57079632679489661923f - t : t;
    const float z = x * x;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(ps * z, x, x);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    s = swap ? cs : sn;
    c = swap ? sn : cs;
}
__device__ __forceinline__ void roots2(float b, float c, float (&r)[3])
{
    r[0] = 0.f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    const float sd = __builtin_sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}
__device__ __forceinline__ void roots3(const float (&m)[9], float (&r)[3])
{
    const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < 1.1920929e-07f) { roots2(c2, c1, r); return; }
    const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = __builtin_sqrtf(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = __builtin_sqrtf(-a_over_3);
    const float theta = sp_atan2_pos(__builtin_sqrtf(-q), half_b) * s_inv3;
    float cos_theta, sin_theta;
    sp_sincos(theta, sin_theta, cos_theta);
    r[0] = c2_over_3 + 2.0f * rho * cos_theta;
    r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    }
    if (r[0] <= 0.0f) roots2(c2, c1, r);
}
__device__ __forceinline__ void eigen33(const float (&mat)[9], float& eigenvalue, float (&v)[3])
{
    float scale = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(mat[i]));
    if (scale <= 1.17549435e-38f) scale = 1.0f;
    float s[9], roots[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = mat[i] / scale;
    roots3(s, roots);
    eigenvalue = roots[0] * scale;
    s[0] -= roots[0]; s[4] -= roots[0]; s[8] -= roots[0];
    const float v1[3] = {s[1] * s[5] - s[2] * s[4], s[2] * s[3] - s[0] * s[5], s[0] * s[4] - s[1] * s[3]};
    const float v2[3] = {s[1] * s[8] - s[2] * s[7], s[2] * s[6] - s[0] * s[8], s[0] * s[7] - s[1] * s[6]};
    const float v3[3] = {s[4] * s[8] - s[5] * s[7], s[5] * s[6] - s[3] * s[8], s[3] * s[7] - s[4] * s[6]};
    const float l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2], l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2],
                l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
    const bool first = l1 >= l2 && l1 >= l3, second = !first && l2 >= l1 && l2 >= l3;
    const float len = first ? l1 : second ? l2 : l3;
    const float sl = __builtin_sqrtf(len);
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = (first ? v1[a] : second ? v2[a] : v3[a]) / sl;
}

// first index q with leaf_key[q] >= key
__device__ __forceinline__ int lower_bound(const unsigned int* __restrict__ leaf_key, int L, unsigned int key)
{
    int lo = 0, hi = L;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (leaf_key[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One WAVE per down-sampled point: k nearest neighbours over the leaf grid, covariance, normal, curvature.
// (One thread per point, the first cut, kept its neighbour list in per-thread arrays -- scratch memory -- and ran with less than one
// wave per SIMD: 18.8 ms for a 63 k-point slab.)  The candidates of radius r -- the leaves in the (2r + 1)^3 cells around the point's
// own: one binary search per x-row of cells, rows dealt to lanes -- go into a list in LDS with their squared distances; the k
// nearest are then picked in ORDER, one per round, as the lexicographic successor of the previous pick under (distance, index) -- a
// strided scan of the list per lane and a wave minimum -- which needs no bookkeeping of what has been taken.  The covariance is
// accumulated in that order (the float sums depend on it), wave-uniformly.  Points that do not find k neighbours within 4 leaf sizes
// (isolated points, a cloud that passed through the grid unfiltered) run the same successor search over ALL leaves.
#define KT_SLICE_MAXC 729   // (2 * 4 + 1)^3 candidates
#define KT_SLICE_RMAX 16
struct slice_pick { float d; int j; };
__device__ __forceinline__ bool slice_less(float da, int ja, float db, int jb) { return da < db || (da == db && ja < jb); }
__device__ __forceinline__ slice_pick slice_wave_min(slice_pick p)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float d2 = __shfl_xor(p.d, off, 64);
        const int j2 = __shfl_xor(p.j, off, 64);
        if (slice_less(d2, j2, p.d, p.j)) { p.d = d2; p.j = j2; }
    }
    return p;
}

__global__ __launch_bounds__(256) void slice_normals(const float* __restrict__ cen, const unsigned int* __restrict__ leaf_key, const unsigned int* __restrict__ leaf_src,
                                                     const Params* __restrict__ prm, int k, const kt_point_xyzrgb* __restrict__ pts,
                                                     kt_point_xyzrgbnormal* __restrict__ out)
{
    __shared__ float s_cd[4][KT_SLICE_MAXC];
    __shared__ int s_cj[4][KT_SLICE_MAXC];
    __shared__ float s_sd[4][KT_SLICE_K_MAX];
    __shared__ int s_sj[4][KT_SLICE_K_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = (int)prm->leaves;
    const Grid g = prm->g;
    const int gridded = prm->gridded;
    float* cd = s_cd[wave]; int* cj = s_cj[wave]; float* sd = s_sd[wave]; int* sj = s_sj[wave];
    // one wave per leaf, a bounded grid striding over the leaves: the number of leaves is known on the device only, and a grid sized for
    // the host's upper bound (the whole extraction buffer in the tracker: 230 k workgroups, nearly all of them empty) is work for the
    // dispatcher on a stream that runs next to the frame path (advisor, round 3)
    for (int q = blockIdx.x * 4 + wave; q < L; q += gridDim.x * 4) {
    const float px = cen[(size_t)q * 6], py = cen[(size_t)q * 6 + 1], pz = cen[(size_t)q * 6 + 2];
    const int kk = min(k, L);
    auto dist2 = [&](int j) -> float {
        const float dx = cen[(size_t)j * 6] - px, dy = cen[(size_t)j * 6 + 1] - py, dz = cen[(size_t)j * 6 + 2] - pz;
        return (dx * dx + dy * dy) + dz * dz;
    };
    // the kk nearest of `nc` listed candidates (or, when nc < 0, of the leaves [jlo, jhi)), in (distance, index) order, into sd / sj;
    // returns how many
    auto select = [&](int nc, int jlo = 0, int jhi = 0) -> int {
        slice_pick prev = {-1.0f, -1};
        int found = 0;
        for (int t = 0; t < kk; ++t) {
            slice_pick best = {3.0e38f, 0x7fffffff};
            if (nc >= 0) {
                for (int i = lane; i < nc; i += 64) {
                    const float d = cd[i]; const int j = cj[i];
                    if (slice_less(prev.d, prev.j, d, j) && slice_less(d, j, best.d, best.j)) { best.d = d; best.j = j; }
                }
            } else {
                for (int j = jlo + lane; j < jhi; j += 64) {
                    const float d = dist2(j);
                    if (slice_less(prev.d, prev.j, d, j) && slice_less(d, j, best.d, best.j)) { best.d = d; best.j = j; }
                }
            }
            best = slice_wave_min(best);
            if (best.j == 0x7fffffff) break;   // the candidates are exhausted
            if (lane == 0) { sd[t] = best.d; sj[t] = best.j; }
            prev = best;
            ++found;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // lane 0's picks are read by every lane of the wave
        __builtin_amdgcn_wave_barrier();
        return found;
    };
    int cnt = 0;
    bool exact = false;
    if (gridded) {
        const unsigned int key = leaf_key[q];
        const int c0 = (int)(key % (unsigned int)g.div_b[0]), c1 = (int)((key / (unsigned int)g.div_b[0]) % (unsigned int)g.div_b[1]),
                  c2 = (int)(key / ((unsigned int)g.div_b[0] * (unsigned int)g.div_b[1]));
        // Every point within r leaf sizes of p (per axis, hence also in Euclidean distance) lies in a cell at most r cells away from
        // p's: a point and its cell index satisfy floor((p + d) / leaf) - floor(p / leaf) <= r for 0 <= d <= r * leaf.  The centroid
        // of a leaf lies inside the leaf up to float rounding; the half-leaf margin below covers that.
        // (k = 20 neighbours on a surface need a disc of radius ~2.5 leaves: r = 2 never proves its answer, so the search starts at 3)
        // radii tried in turn: 3 (2 for small k), 4, then 8 and 16 -- at the edge of a 15-voxel shift slab a 20-neighbour disc is cut in
        // half (a quarter of a slab's points), and the missing neighbours lie further along the strip; a radius whose cells hold more
        // than KT_SLICE_MAXC leaves (a dense cloud) is given up for the windowed search below
        bool overflow = false;
        for (int r = (kk > 7 ? 3 : 2); r <= KT_SLICE_RMAX && !exact && !overflow; r = r < 4 ? r + 1 : 2 * r) {
            const int side = 2 * r + 1, rows = side * side;
            int nc = 0;   // wave-uniform
            for (int row0 = 0; row0 < rows && !overflow; row0 += 64) {
                const int row = row0 + lane;
                int lo = 0, n_here = 0;
                if (row < rows) {
                    const int y = c1 + row % side - r, z = c2 + row / side - r;
                    if (y >= 0 && y < g.div_b[1] && z >= 0 && z < g.div_b[2]) {
                        // the 2r + 1 cells of this x-row have consecutive keys: one binary search for each end
                        const int x0 = max(0, c0 - r), x1 = min(g.div_b[0] - 1, c0 + r);
                        const unsigned int k0 = (unsigned int)(x0 + y * g.div_b[0] + z * g.div_b[0] * g.div_b[1]), k1 = k0 + (unsigned int)(x1 - x0);
                        lo = lower_bound(leaf_key, L, k0);
                        int hi = lo;
                        if (r <= 4) { while (hi < L && leaf_key[hi] <= k1) ++hi; }   // a short scan ...
                        else hi = lower_bound(leaf_key, L, k1 + 1u);                 // ... or a second search when the row is long
                        n_here = hi - lo;
                    }
                }
                // exclusive prefix of the rows' counts over the wave: where this lane's candidates go
                int incl = n_here;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += up;
                }
                const int base = nc + incl - n_here;
                const int total = nc + __shfl(incl, 63, 64);
                if (total > KT_SLICE_MAXC) { overflow = true; break; }   // (wave-uniform)
                for (int i = 0; i < n_here; ++i) { cd[base + i] = dist2(lo + i); cj[base + i] = lo + i; }
                nc = total;
            }
            if (overflow) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the list is written by the lanes that own the rows, scanned by all
            __builtin_amdgcn_wave_barrier();
            cnt = select(nc);
            const float reach = ((float)r - 0.5f) * g.leaf;
            exact = cnt == kk && sd[kk - 1] <= reach * reach;   // (sd: written by lane 0, read by every lane of the same wave: in order within a wave)
        }
    }
    // Points the cell search cannot settle used to fall back on ALL leaves -- 20 passes over 50 000 of them, most of the stage's time
    // until round 4.  What is left after the larger radii above: the keys are sorted z-major, so the leaves within R cells of the
    // point's z are one contiguous run of the list: the same successor search over that run, R = 8, 16, ... until the k-th neighbour is
    // provably the k-th nearest (every point within (R - 1/2) leaf sizes lies inside the window) or the window is the whole list.
    if (!exact && gridded) {
        const int c2 = (int)(leaf_key[q] / ((unsigned int)g.div_b[0] * (unsigned int)g.div_b[1]));
        const unsigned int plane_cells = (unsigned int)g.div_b[0] * (unsigned int)g.div_b[1];
        for (int R = 8; !exact; R *= 2) {
            const int zlo = max(0, c2 - R), zhi = min(g.div_b[2] - 1, c2 + R);
            const bool all = zlo == 0 && zhi == g.div_b[2] - 1;
            const int jlo = all ? 0 : lower_bound(leaf_key, L, (unsigned int)zlo * plane_cells);
            const int jhi = (all || zhi == g.div_b[2] - 1) ? L : lower_bound(leaf_key, L, (unsigned int)(zhi + 1) * plane_cells);
            cnt = select(-1, jlo, jhi);
            const float reach = ((float)R - 0.5f) * g.leaf;
            exact = all || (cnt == kk && sd[kk - 1] <= reach * reach);
        }
    }
    if (!exact) cnt = select(-1, 0, L);
    kt_point_xyzrgbnormal o;
    o.x = px; o.y = py; o.z = pz; o.pad0 = 1.0f;
    o.pad1 = 0.0f; o.pad2[0] = 0.0f; o.pad2[1] = 0.0f;
    // VoxelGrid: r, g, b = static_cast<uint8_t> of the float means; the packed rgb has a zero alpha byte -- except in the "leaf size
    // too small" case, where `output = *input_` keeps every point as it is, its weight byte included
    o.r = (unsigned char)cen[(size_t)q * 6 + 3]; o.g = (unsigned char)cen[(size_t)q * 6 + 4]; o.b = (unsigned char)cen[(size_t)q * 6 + 5];
    o.a = gridded ? (unsigned char)0 : pts[leaf_src[q]].a;
    if (cnt < 3) {   // NormalEstimation: fewer than 3 neighbours -> NaN normal and curvature
        o.normal_x = o.normal_y = o.normal_z = o.curvature = __builtin_nanf("");
        if (lane == 0) out[q] = o;
        continue;
    }
    // lane t fetches pick t (one round of loads for all of them); the sums then run over the picks in order, every lane alike
    const int jt = sj[min(lane, cnt - 1)];
    const float xt = cen[(size_t)jt * 6], yt = cen[(size_t)jt * 6 + 1], zt = cen[(size_t)jt * 6 + 2];
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < cnt; ++t) {
        const float x = __shfl(xt, t, 64), y = __shfl(yt, t, 64), z = __shfl(zt, t, 64);
        acc[0] += x * x; acc[1] += x * y; acc[2] += x * z; acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
        acc[6] += x; acc[7] += y; acc[8] += z;
    }
    const float inv_cnt = 1.0f / (float)cnt;   // computeMeanAndCovarianceMatrix: `accu /= static_cast<Scalar>(point_count)`, Eigen 3.2 (see slice_centroids)
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] *= inv_cnt;
    float cov[9];
    cov[0] = acc[0] - acc[6] * acc[6]; cov[1] = acc[1] - acc[6] * acc[7]; cov[2] = acc[2] - acc[6] * acc[8];
    cov[4] = acc[3] - acc[7] * acc[7]; cov[5] = acc[4] - acc[7] * acc[8]; cov[8] = acc[5] - acc[8] * acc[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, nv[3];
    eigen33(cov, ev, nv);
    const float eig_sum = cov[0] + cov[4] + cov[8];
    o.curvature = eig_sum != 0 ? fabsf(ev / eig_sum) : 0;
    const float cos_theta = ((0.0f - px) * nv[0] + (0.0f - py) * nv[1]) + (0.0f - pz) * nv[2];   // viewpoint = sensor origin
    if (cos_theta < 0) { nv[0] *= -1; nv[1] *= -1; nv[2] *= -1; }
    o.normal_x = nv[0]; o.normal_y = nv[1]; o.normal_z = nv[2];
    if (lane == 0) out[q] = o;
    }
}

}  // namespace

// ---- workspace + entry points ---------------------------------------------------------------------------------------------------
struct kt_slice_ws {
    kt_ctx* ctx;
    hipStream_t stream; bool own_stream;
    size_t cap;
    unsigned int *keys[2], *src[2], *head, *leafof, *leaf_key, *leaf_src, *n_dev;
    float* cen;
    BBox* box;
    Params* prm;
    void* tmp; size_t tmp_bytes;
    kt_point_xyzrgb* in;              // staging for the host-array entry point
    kt_point_xyzrgbnormal* out;       // device output of the last call (cap points)
    unsigned int* leaves_host;        // pinned: the output count, written behind the last launch
};

extern "C" int kt_slice_ws_destroy(kt_slice_ws* w)
{
    if (!w) return KT_OK;
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    for (int k = 0; k < 2; ++k) { (void)hipFree(w->keys[k]); (void)hipFree(w->src[k]); }
    (void)hipFree(w->head); (void)hipFree(w->leafof); (void)hipFree(w->leaf_key); (void)hipFree(w->leaf_src); (void)hipFree(w->n_dev);
    (void)hipFree(w->cen); (void)hipFree(w->box); (void)hipFree(w->prm); (void)hipFree(w->tmp); (void)hipFree(w->in); (void)hipFree(w->out);
    (void)hipHostFree(w->leaves_host);
    if (w->own_stream && w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
    return KT_OK;
}

// capacity = the largest number of input points a call may bring; stream = the stream the stage runs on (null: one of its own,
// so that a backend thread's slices never queue behind -- or in front of -- a frame)
extern "C" int kt_slice_ws_create(kt_ctx* c, size_t capacity, void* hip_stream, kt_slice_ws** out)
{
    KT_ARG(c && out && capacity > 0 && capacity < (1u << 30));
    kt_slice_ws* w = new kt_slice_ws();
    memset(w, 0, sizeof(*w));
    w->ctx = c; w->cap = capacity;
    int s = KT_OK;
    auto A = [&](void** p, size_t bytes) { if (s == KT_OK) s = kt_check(hipMalloc(p, bytes), "hipMalloc", __FILE__, __LINE__); };
    if (hip_stream) w->stream = (hipStream_t)hip_stream;
    else { s = kt_check(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking), "hipStreamCreateWithFlags", __FILE__, __LINE__); w->own_stream = s == KT_OK; }
    for (int k = 0; k < 2; ++k) { A((void**)&w->keys[k], capacity * 4); A((void**)&w->src[k], capacity * 4); }
    A((void**)&w->head, capacity * 4); A((void**)&w->leafof, capacity * 4); A((void**)&w->leaf_key, capacity * 4); A((void**)&w->leaf_src, capacity * 4);
    A((void**)&w->n_dev, 4); A((void**)&w->cen, capacity * 6 * sizeof(float)); A((void**)&w->box, sizeof(BBox) * KT_SLICE_BOXES); A((void**)&w->prm, sizeof(Params));
    A((void**)&w->in, capacity * sizeof(kt_point_xyzrgb)); A((void**)&w->out, capacity * sizeof(kt_point_xyzrgbnormal));
    if (s == KT_OK) {
        size_t a = 0, b = 0;
        s = kt_check(rocprim::radix_sort_pairs(nullptr, a, w->keys[0], w->keys[1], w->src[0], w->src[1], (unsigned int)capacity, 0, 32, w->stream), "rocprim::radix_sort_pairs", __FILE__, __LINE__);
        if (s == KT_OK) s = kt_check(rocprim::inclusive_scan(nullptr, b, w->head, w->leafof, capacity, rocprim::plus<unsigned int>(), w->stream), "rocprim::inclusive_scan", __FILE__, __LINE__);
        w->tmp_bytes = a > b ? a : b;
        A(&w->tmp, w->tmp_bytes ? w->tmp_bytes : 16);
    }
    if (s == KT_OK) s = kt_check(hipHostMalloc((void**)&w->leaves_host, sizeof(unsigned int), hipHostMallocDefault), "hipHostMalloc", __FILE__, __LINE__);
    if (s != KT_OK) { (void)kt_slice_ws_destroy(w); return s; }
    *out = w;
    return KT_OK;
}

extern "C" void* kt_slice_ws_stream(kt_slice_ws* w) { return w ? (void*)w->stream : nullptr; }
extern "C" const kt_point_xyzrgbnormal* kt_slice_ws_output(kt_slice_ws* w) { return w ? w->out : nullptr; }

// The stage on device-resident points: points_dev[0 .. *n_dev) (n_max = an upper bound of *n_dev known to the host, <= the workspace's
// capacity; n_dev itself is read on the device, e.g. the extraction kernel's own counter).  Everything is enqueued on the workspace's
// stream and nothing waits: the result is kt_slice_ws_output(ws)[0 .. *count) once that stream has been synchronised, *count =
// the pinned word the last copy fills (kt_slice_ws_count).
extern "C" int kt_slice_process_device(kt_slice_ws* w, const kt_point_xyzrgb* points_dev, const unsigned int* n_dev, size_t n_max, int weight_cull, float leaf, int k)
{
    KT_ARG(w && points_dev && n_dev && n_max > 0 && n_max <= w->cap && leaf > 0 && k >= 1 && k <= KT_SLICE_K_MAX);
    hipStream_t st = w->stream;
    const int nm = (int)n_max, nb = kt_div_up(nm, 256);
    hipLaunchKernelGGL(slice_bbox, dim3(KT_SLICE_BOXES), dim3(256), 0, st, points_dev, n_dev, nm, weight_cull, w->box);
    hipLaunchKernelGGL(slice_grid, dim3(1), dim3(64), 0, st, w->box, leaf, w->prm);
    hipLaunchKernelGGL(slice_keys, dim3(nb), dim3(256), 0, st, points_dev, n_dev, nm, weight_cull, w->prm, w->keys[0], w->src[0]);
    KT_LAUNCH_CHECK();
    size_t tb = w->tmp_bytes;
    KT_HIP(rocprim::radix_sort_pairs(w->tmp, tb, w->keys[0], w->keys[1], w->src[0], w->src[1], (unsigned int)nm, 0, 32, st));   // stable
    hipLaunchKernelGGL(slice_heads, dim3(nb), dim3(256), 0, st, w->keys[1], nm, w->head);
    KT_LAUNCH_CHECK();
    tb = w->tmp_bytes;
    KT_HIP(rocprim::inclusive_scan(w->tmp, tb, w->head, w->leafof, (size_t)nm, rocprim::plus<unsigned int>(), st));
    hipLaunchKernelGGL(slice_centroids, dim3(nb), dim3(256), 0, st, points_dev, w->keys[1], w->src[1], w->head, w->leafof, nm, w->cen, w->leaf_key, w->leaf_src, w->prm);
    // NormalEstimation (kNN) + concatenateFields: one wave per leaf, a bounded grid striding over the leaves
    hipLaunchKernelGGL(slice_normals, dim3(kt_div_up(nm, 4) < 4096 ? kt_div_up(nm, 4) : 4096), dim3(256), 0, st, w->cen, w->leaf_key, w->leaf_src, w->prm, k, points_dev, w->out);
    KT_LAUNCH_CHECK();
    KT_HIP(hipMemcpyAsync(w->leaves_host, &w->prm->leaves, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    return KT_OK;
}

// copy `*n_dev` (clamped to cap) items of `item16` 16-byte words each, and the clamped count itself: how a variable-length device
// result reaches pinned host memory without the host knowing its length (the tracker's slice download)
__global__ __launch_bounds__(256) void kt_copy_counted_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const unsigned int* __restrict__ n_dev,
                                                              unsigned int cap, int item16, unsigned int* __restrict__ count_out)
{
    const unsigned int n = min(*n_dev, cap);
    const size_t total = (size_t)n * item16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = n;
}
int kt_copy_counted(hipStream_t st, const void* src, void* dst, const unsigned int* n_dev, unsigned int cap, int item_bytes, unsigned int* count_out)
{
    hipLaunchKernelGGL(kt_copy_counted_kernel, dim3(512), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n_dev, cap, item_bytes / 16, count_out);
    KT_LAUNCH_CHECK();
    return KT_OK;
}
const unsigned int* kt_slice_ws_leaves_dev(kt_slice_ws* w) { return &w->prm->leaves; }

extern "C" int kt_slice_ws_count(kt_slice_ws* w, size_t* n_out)
{
    KT_ARG(w && n_out);
    KT_HIP(hipStreamSynchronize(w->stream));
    *n_out = (size_t)*w->leaves_host;
    return KT_OK;
}

// the host-array form (a CloudSlice's cloud in, its processedCloud out): upload, the device stage, download -- on a workspace the
// context keeps (grown when a larger slice arrives) and on the context's stream
extern "C" int kt_slice_process(kt_ctx* c, const kt_point_xyzrgb* points_host, size_t n_in, int weight_cull, float leaf, int k,
                                kt_point_xyzrgbnormal* out_host, size_t* n_out)
{
    KT_ARG(c && n_out && (n_in == 0 || (points_host && out_host)) && leaf > 0 && k >= 1 && k <= KT_SLICE_K_MAX && n_in < (1u << 30));
    *n_out = 0;
    if (n_in == 0) return KT_OK;
    kt_slice_ws* w = (kt_slice_ws*)c->slice_ws;
    if (!w || w->cap < n_in || w->stream != c->stream) {
        if (w) (void)kt_slice_ws_destroy(w);
        c->slice_ws = nullptr;
        size_t cap = 1 << 16;
        while (cap < n_in) cap <<= 1;
        KT_TRY(kt_slice_ws_create(c, cap, (void*)c->stream, &w));
        c->slice_ws = w;
    }
    const unsigned int n32 = (unsigned int)n_in;
    KT_HIP(hipMemcpyAsync(w->in, points_host, n_in * sizeof(kt_point_xyzrgb), hipMemcpyHostToDevice, w->stream));
    KT_HIP(hipMemcpyAsync(w->n_dev, &n32, sizeof(n32), hipMemcpyHostToDevice, w->stream));
    KT_TRY(kt_slice_process_device(w, w->in, w->n_dev, n_in, weight_cull, leaf, k));
    size_t L = 0;
    KT_TRY(kt_slice_ws_count(w, &L));
    if (L) {
        KT_HIP(hipMemcpyAsync(out_host, w->out, L * sizeof(kt_point_xyzrgbnormal), hipMemcpyDeviceToHost, w->stream));
        KT_HIP(hipStreamSynchronize(w->stream));
    }
    *n_out = L;
    return KT_OK;
}
