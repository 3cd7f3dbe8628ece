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
// Shape on the GPU: cull = flags + scan + scatter; leaf keys; ONE stable radix sort of (key, index) (rocPRIM through hipCUB: the sort is
// library plumbing, everything arithmetic is written here); one thread per leaf walks its short run in order; after the grid every
// point owns a distinct leaf cell, so the k nearest neighbours of a point are found by binary-searching the sorted leaf keys of the
// (2r + 1)^3 cells around its own cell -- all points within r leaf sizes are in there -- with r growing until the k-th neighbour is
// provably the k-th nearest (distance <= r * leaf), then covariance, smallest eigenvector and curvature in the same thread.
#include "kt_internal.hpp"

#include <hipcub/hipcub.hpp>

#define KT_SLICE_K_MAX 64

namespace {

struct BBox { float mn[3], mx[3]; };

__global__ __launch_bounds__(256) void slice_cull_flags(const kt_point_xyzrgb* __restrict__ in, int n, int weight_cull, unsigned int* __restrict__ keep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keep[i] = (!(weight_cull > 0) || (int)in[i].a >= weight_cull) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void slice_cull_scatter(const kt_point_xyzrgb* __restrict__ in, int n, const unsigned int* __restrict__ keep,
                                                          const unsigned int* __restrict__ pos, kt_point_xyzrgb* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && keep[i]) out[pos[i]] = in[i];
}

// getMinMax3D: per-workgroup partial boxes, folded by one more launch of the same kernel over the partials
__global__ __launch_bounds__(256) void slice_bbox(const float* __restrict__ xyz, int stride_floats, int n, BBox* __restrict__ partial)
{
    __shared__ BBox sh[4];
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[(size_t)i * stride_floats + a], w = xyz[(size_t)i * stride_floats + (stride_floats == 6 ? 3 : 0) + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], stride_floats == 6 ? w : v);
        }
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

struct Grid { int min_b[3], div_b[3]; float inv_leaf, leaf; };

__global__ __launch_bounds__(256) void slice_keys(const kt_point_xyzrgb* __restrict__ pts, int m, Grid g, unsigned int* __restrict__ keys,
                                                  unsigned int* __restrict__ src)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    // voxel_grid.hpp: ijk = static_cast<int>(floor(p * inverse_leaf_size) - static_cast<float>(min_b)); idx = ijk . divb_mul
    const int i0 = (int)(__builtin_floorf(pts[i].x * g.inv_leaf) - (float)g.min_b[0]);
    const int i1 = (int)(__builtin_floorf(pts[i].y * g.inv_leaf) - (float)g.min_b[1]);
    const int i2 = (int)(__builtin_floorf(pts[i].z * g.inv_leaf) - (float)g.min_b[2]);
    keys[i] = (unsigned int)(i0 + i1 * g.div_b[0] + i2 * g.div_b[0] * g.div_b[1]);
    src[i] = (unsigned int)i;
}

__global__ __launch_bounds__(256) void slice_heads(const unsigned int* __restrict__ keys, int m, unsigned int* __restrict__ head)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// one thread per leaf (= per run of equal keys in the sorted order): the centroid of x, y, z and of r, g, b as floats, summed in run order
__global__ __launch_bounds__(256) void slice_centroids(const kt_point_xyzrgb* __restrict__ pts, const unsigned int* __restrict__ keys,
                                                       const unsigned int* __restrict__ src, const unsigned int* __restrict__ head,
                                                       const unsigned int* __restrict__ leaf_of, int m, float* __restrict__ cen,
                                                       unsigned int* __restrict__ leaf_key)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m || !head[i]) return;
    const unsigned int key = keys[i];
    float acc[6] = {0, 0, 0, 0, 0, 0};
    int j = i;
    for (; j < m && keys[j] == key; ++j) {
        const kt_point_xyzrgb p = pts[src[j]];
        acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += (float)p.r; acc[4] += (float)p.g; acc[5] += (float)p.b;
    }
    // `centroid /= static_cast<float>(count)` on an Eigen::VectorXf: Eigen 3.2 (the reference's, README.md:14-31) evaluates a
    // floating-point `/= s` as a multiplication by Scalar(1) / s (SelfCwiseBinaryOp.h; true division only from 3.3 on)
    const float inv_cnt = 1.0f / (float)(j - i);
    const unsigned int q = leaf_of[i] - 1u;   // inclusive scan of the head flags
#pragma unroll
    for (int a = 0; a < 6; ++a) cen[(size_t)q * 6 + a] = acc[a] * inv_cnt;
    leaf_key[q] = key;
}

// ---- pcl::eigen33 / computeRoots (common/impl/eigen.hpp), float ----
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
    const float theta = atan2f(__builtin_sqrtf(-q), half_b) * s_inv3;
    const float cos_theta = cosf(theta), sin_theta = sinf(theta);
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

// one thread per down-sampled point: kNN over the leaf grid, covariance, normal, curvature
__global__ __launch_bounds__(128) void slice_normals(const float* __restrict__ cen, const unsigned int* __restrict__ leaf_key, int L, Grid g, int k,
                                                     int gridded, const kt_point_xyzrgb* __restrict__ pts, kt_point_xyzrgbnormal* __restrict__ out)
{
    const int q = blockIdx.x * 128 + threadIdx.x;
    if (q >= L) return;
    const float px = cen[(size_t)q * 6], py = cen[(size_t)q * 6 + 1], pz = cen[(size_t)q * 6 + 2];
    const int kk = min(k, L);
    float bd[KT_SLICE_K_MAX];
    int bi[KT_SLICE_K_MAX];
    int cnt = 0;
    auto offer = [&](int j) {
        const float dx = cen[(size_t)j * 6] - px, dy = cen[(size_t)j * 6 + 1] - py, dz = cen[(size_t)j * 6 + 2] - pz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        // ordered by (distance, index): candidates do not arrive in index order here, so ties compare the index explicitly
        if (cnt == kk && !(d < bd[cnt - 1] || (d == bd[cnt - 1] && j < bi[cnt - 1]))) return;
        int pos = cnt < kk ? cnt : kk - 1;
        while (pos > 0 && (d < bd[pos - 1] || (d == bd[pos - 1] && j < bi[pos - 1]))) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
        bd[pos] = d; bi[pos] = j;
        if (cnt < kk) ++cnt;
    };
    bool exact = false;
    if (gridded) {
        const unsigned int key = leaf_key[q];
        const int c0 = (int)(key % (unsigned int)g.div_b[0]), c1 = (int)((key / (unsigned int)g.div_b[0]) % (unsigned int)g.div_b[1]),
                  c2 = (int)(key / ((unsigned int)g.div_b[0] * (unsigned int)g.div_b[1]));
        // Every point within r leaf sizes of p (per axis, hence also in Euclidean distance) lies in a cell at most r cells away from
        // p's: a point and its cell index satisfy floor((p + d) / leaf) - floor(p / leaf) <= r for 0 <= d <= r * leaf.  The centroid
        // of a leaf lies inside the leaf up to float rounding; the half-leaf margin below covers that.
        for (int r = 2; r <= 6 && !exact; ++r) {
            cnt = 0;
            for (int dz = -r; dz <= r; ++dz) {
                const int z = c2 + dz;
                if (z < 0 || z >= g.div_b[2]) continue;
                for (int dy = -r; dy <= r; ++dy) {
                    const int y = c1 + dy;
                    if (y < 0 || y >= g.div_b[1]) continue;
                    // the 2r + 1 cells of this x-row have consecutive keys: one binary search, then a short scan
                    const int x0 = max(0, c0 - r), x1 = min(g.div_b[0] - 1, c0 + r);
                    const unsigned int k0 = (unsigned int)(x0 + y * g.div_b[0] + z * g.div_b[0] * g.div_b[1]), k1 = k0 + (unsigned int)(x1 - x0);
                    for (int j = lower_bound(leaf_key, L, k0); j < L && leaf_key[j] <= k1; ++j) offer(j);
                }
            }
            const float reach = ((float)r - 0.5f) * g.leaf;
            exact = cnt == kk && bd[cnt - 1] <= reach * reach;
        }
    }
    if (!exact) {   // isolated points, a cloud that passed through the grid unfiltered: every point is a candidate
        cnt = 0;
        for (int j = 0; j < L; ++j) offer(j);
    }
    kt_point_xyzrgbnormal o;
    o.x = px; o.y = py; o.z = pz; o.pad0 = 1.0f;
    o.pad1 = 0.0f; o.pad2[0] = 0.0f; o.pad2[1] = 0.0f;
    // VoxelGrid: r, g, b = static_cast<uint8_t> of the float means; the packed rgb has a zero alpha byte -- except in the "leaf size
    // too small" case, where `output = *input_` keeps every point as it is, its weight byte included
    o.r = (unsigned char)cen[(size_t)q * 6 + 3]; o.g = (unsigned char)cen[(size_t)q * 6 + 4]; o.b = (unsigned char)cen[(size_t)q * 6 + 5];
    o.a = gridded ? (unsigned char)0 : pts[q].a;
    if (cnt < 3) {   // NormalEstimation: fewer than 3 neighbours -> NaN normal and curvature
        o.normal_x = o.normal_y = o.normal_z = o.curvature = __builtin_nanf("");
        out[q] = o;
        return;
    }
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < cnt; ++t) {
        const float x = cen[(size_t)bi[t] * 6], y = cen[(size_t)bi[t] * 6 + 1], z = cen[(size_t)bi[t] * 6 + 2];
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
    out[q] = o;
}

// pass-through rows for the "leaf size too small" case: centroid records straight from the points
__global__ __launch_bounds__(256) void slice_passthrough(const kt_point_xyzrgb* __restrict__ pts, int m, float* __restrict__ cen, unsigned int* __restrict__ leaf_key)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    cen[(size_t)i * 6] = pts[i].x; cen[(size_t)i * 6 + 1] = pts[i].y; cen[(size_t)i * 6 + 2] = pts[i].z;
    cen[(size_t)i * 6 + 3] = (float)pts[i].r; cen[(size_t)i * 6 + 4] = (float)pts[i].g; cen[(size_t)i * 6 + 5] = (float)pts[i].b;
    leaf_key[i] = (unsigned int)i;
}

struct DevBufs {   // freed on every exit path
    void* p[16];
    int n;
    DevBufs() : n(0) {}
    ~DevBufs() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
    template <class T> int alloc(T** out, size_t count)
    {
        void* q = nullptr;
        if (hipMalloc(&q, (count ? count : 1) * sizeof(T)) != hipSuccess) { kt_set_error("kt_slice_process: out of device memory"); return KT_ERR_NOMEM; }
        p[n++] = q;
        *out = (T*)q;
        return KT_OK;
    }
};

}  // namespace

extern "C" int kt_slice_process(kt_ctx* c, const kt_point_xyzrgb* points_host, size_t n_in, int weight_cull, float leaf, int k,
                                kt_point_xyzrgbnormal* out_host, size_t* n_out)
{
    KT_ARG(c && n_out && (n_in == 0 || (points_host && out_host)) && leaf > 0 && k >= 1 && k <= KT_SLICE_K_MAX && n_in < (1u << 30));
    *n_out = 0;
    if (n_in == 0) return KT_OK;
    const int n = (int)n_in;
    hipStream_t st = c->stream;
    DevBufs b;
    kt_point_xyzrgb *d_in, *d_pts;
    unsigned int *d_keep, *d_pos, *d_keys, *d_keys2, *d_src, *d_src2, *d_head, *d_leafof, *d_leafkey;
    float* d_cen;
    BBox* d_box;
    kt_point_xyzrgbnormal* d_out;
    KT_TRY(b.alloc(&d_in, n)); KT_TRY(b.alloc(&d_pts, n)); KT_TRY(b.alloc(&d_keep, n)); KT_TRY(b.alloc(&d_pos, n));
    KT_HIP(hipMemcpyAsync(d_in, points_host, (size_t)n * sizeof(kt_point_xyzrgb), hipMemcpyHostToDevice, st));
    const int nb = kt_div_up(n, 256);
    // ---- weight cull (CloudSliceProcessor.cpp:99-117), order preserved ----
    hipLaunchKernelGGL(slice_cull_flags, dim3(nb), dim3(256), 0, st, d_in, n, weight_cull, d_keep);
    KT_LAUNCH_CHECK();
    size_t tmp_bytes = 0, need = 0;
    KT_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_keep, d_pos, n, st));
    tmp_bytes = need;
    KT_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const unsigned int*)nullptr,
                                              (unsigned int*)nullptr, n, 0, 32, st));
    tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
    KT_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, need, d_keep, d_pos, n, st));
    tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
    unsigned char* d_tmp;
    KT_TRY(b.alloc(&d_tmp, tmp_bytes));
    KT_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_keep, d_pos, n, st));
    hipLaunchKernelGGL(slice_cull_scatter, dim3(nb), dim3(256), 0, st, d_in, n, d_keep, d_pos, d_pts);
    KT_LAUNCH_CHECK();
    unsigned int last_pos = 0, last_keep = 0;
    KT_HIP(hipMemcpyAsync(&last_pos, d_pos + (n - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    KT_HIP(hipMemcpyAsync(&last_keep, d_keep + (n - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    KT_HIP(hipStreamSynchronize(st));
    const int m = (int)(last_pos + last_keep);
    if (m == 0) return KT_OK;
    const int mb = kt_div_up(m, 256);
    // ---- VoxelGrid::applyFilter ----
    const int boxes = min(256, mb);
    KT_TRY(b.alloc(&d_box, boxes + 1));
    hipLaunchKernelGGL(slice_bbox, dim3(boxes), dim3(256), 0, st, &d_pts->x, 8, m, d_box);
    KT_LAUNCH_CHECK();
    hipLaunchKernelGGL(slice_bbox, dim3(1), dim3(256), 0, st, &d_box->mn[0], 6, boxes, d_box + boxes);
    KT_LAUNCH_CHECK();
    BBox box;
    KT_HIP(hipMemcpyAsync(&box, d_box + boxes, sizeof(BBox), hipMemcpyDeviceToHost, st));
    KT_HIP(hipStreamSynchronize(st));
    Grid g;
    g.leaf = leaf;
    g.inv_leaf = 1.0f / leaf;
    int max_b[3];
    for (int a = 0; a < 3; ++a) {
        g.min_b[a] = (int)floorf(box.mn[a] * g.inv_leaf);
        max_b[a] = (int)floorf(box.mx[a] * g.inv_leaf);
        g.div_b[a] = max_b[a] - g.min_b[a] + 1;
    }
    KT_TRY(b.alloc(&d_cen, (size_t)m * 6)); KT_TRY(b.alloc(&d_leafkey, m)); KT_TRY(b.alloc(&d_out, m));
    int L = 0, gridded = 1;
    // voxel_grid.hpp: dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inverse_leaf_size_[0]) + 1, ...; dx * dy * dz > INT32_MAX
    if (((long long)((box.mx[0] - box.mn[0]) * g.inv_leaf) + 1) * ((long long)((box.mx[1] - box.mn[1]) * g.inv_leaf) + 1) *
            ((long long)((box.mx[2] - box.mn[2]) * g.inv_leaf) + 1) > 2147483647LL) {
        // "Leaf size is too small for the input dataset. Integer indices would overflow.": PCL passes the cloud through unfiltered
        hipLaunchKernelGGL(slice_passthrough, dim3(mb), dim3(256), 0, st, d_pts, m, d_cen, d_leafkey);
        KT_LAUNCH_CHECK();
        L = m;
        gridded = 0;
    } else {
        KT_TRY(b.alloc(&d_keys, m)); KT_TRY(b.alloc(&d_keys2, m)); KT_TRY(b.alloc(&d_src, m)); KT_TRY(b.alloc(&d_src2, m));
        KT_TRY(b.alloc(&d_head, m)); KT_TRY(b.alloc(&d_leafof, m));
        hipLaunchKernelGGL(slice_keys, dim3(mb), dim3(256), 0, st, d_pts, m, g, d_keys, d_src);
        KT_LAUNCH_CHECK();
        KT_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_src, d_src2, m, 0, 32, st));   // stable
        hipLaunchKernelGGL(slice_heads, dim3(mb), dim3(256), 0, st, d_keys2, m, d_head);
        KT_LAUNCH_CHECK();
        KT_HIP(hipcub::DeviceScan::InclusiveSum(d_tmp, tmp_bytes, d_head, d_leafof, m, st));
        hipLaunchKernelGGL(slice_centroids, dim3(mb), dim3(256), 0, st, d_pts, d_keys2, d_src2, d_head, d_leafof, m, d_cen, d_leafkey);
        KT_LAUNCH_CHECK();
        unsigned int leaves = 0;
        KT_HIP(hipMemcpyAsync(&leaves, d_leafof + (m - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        KT_HIP(hipStreamSynchronize(st));
        L = (int)leaves;
    }
    // ---- NormalEstimation (kNN) + concatenateFields ----
    hipLaunchKernelGGL(slice_normals, dim3(kt_div_up(L, 128)), dim3(128), 0, st, d_cen, d_leafkey, L, g, k, gridded, d_pts, d_out);
    KT_LAUNCH_CHECK();
    KT_HIP(hipMemcpyAsync(out_host, d_out, (size_t)L * sizeof(kt_point_xyzrgbnormal), hipMemcpyDeviceToHost, st));
    KT_HIP(hipStreamSynchronize(st));
    *n_out = (size_t)L;
    return KT_OK;
}
