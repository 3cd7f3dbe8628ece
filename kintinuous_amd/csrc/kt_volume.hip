// kt_volume.hip -- cyclical TSDF volume kernels for gfx950: integrate (a11), raycast (a12),
// slab clear (a14), cloud-slice extraction (a15), volume init.  Hand-written wave64 HIP; no MFMA
// (there is no dense contraction on this path).  Reference: src/frontend/cuda/tsdf_volume.cu,
// ray_caster.cu, extract.cu (file:line cited per kernel).  Results are bit-identical to the oracle's
// restatement of those files: same operation order, explicit fmaf sites, IEEE div/sqrt.
//
// HBM layout: tsdf = int16[N^3] (x fastest), colour+weight = uchar4[N^3]; storage index is the logical
// index rotated by voxel_wrap (tsdf_volume.cu:612).  Kernels are laid out so that the 64 lanes of a
// wave own 64 consecutive STORAGE x of one (y, z) line: 128 B of tsdf + 256 B of colour per access.
#include "kt_internal.hpp"
#include "kt_pyramid.hpp"

#include <stdlib.h>
#include <string.h>

thread_local kt_event_hook kt_tsdf23_hook = {{nullptr, nullptr}, false};

// ================================================================================================
// init  (initVolume / initColorVolume, tsdf_volume.cu:56-87, 450-479): pack_tsdf(0) == 0, uchar4(0)
// ================================================================================================
extern "C" int kt_init_volume(kt_ctx* c, int16_t* volume, int N)
{
    KT_ARG(c && volume && N > 0);
    KT_HIP(hipMemsetAsync(volume, 0, (size_t)N * N * N * sizeof(int16_t), c->stream));
    return KT_OK;
}
extern "C" int kt_init_color_volume(kt_ctx* c, uint8_t* cv, int N)
{
    KT_ARG(c && cv && N > 0);
    KT_HIP(hipMemsetAsync(cv, 0, (size_t)N * N * N * 4, c->stream));
    return KT_OK;
}

// ================================================================================================
// a11  integrateTsdfVolume = scaleDepth + tsdf23      tsdf_volume.cu:490-674
// ================================================================================================
// Per-pixel record consumed by the voxel kernel: everything tsdf23 gathers per projected pixel
// (scaled depth with the no-colour sign flag, the colour weight derived from |n_z|, rgb, normal-valid)
// packed into ONE 16-byte gather instead of 8 scalar gathers.  All fields are computed with exactly
// the reference's expressions, only earlier.
// edge of the bricks whose "holds a negative tsdf" flags steer the raycast's empty-space hops (kt_raycast_kernel)
#define KT_BRICK_LOG2 5   // 32^3: 16^3 costs 32 KB of LDS staging and twice the hops (97 us), 64^3 leaves a thicker flagged shell (68 us vs 66)
#define KT_BRICK (1 << KT_BRICK_LOG2)
size_t kt_brick_count(int N) { const size_t nb = (size_t)(N + KT_BRICK - 1) / KT_BRICK; return nb * nb * nb; }

struct kt_integrate_tables {  // incremental z walk of tsdf23 (quirk A.17), identical for every column
    float* vgz;      // v_g_z after z increments
    float* zs;       // z_scaled after z increments
};

// One workgroup = a 32-pixel-wide strip of `passes` x 8 rows.  With tile_raw set (the tracker's read-ahead path) passes x 8 is the tile edge T of
// the depth-range map (8, 16 or 32: kt_dpt_log2), so a workgroup covers 32 / T whole tiles and leaves their raw maxima of |scaled depth| --
// round 4 spent a launch of its own on them (kt_tile_max_kernel: 5 us of launch latency on the read-ahead stream).
// Round 6: the 7 x 7 window of the angleColor test (scaleDepth, tsdf_volume.cu:490-527) is read from an LDS copy of the strip with its 3-pixel rim
// instead of 49 global loads per pixel (13 us alone at 640x480 -> 4).  The window is clipped and upper-exclusive like the bilateral filter's (quirk
// A.5): cx runs over [max(x - 3, 0), min(x + 4, cols - 1)), so the last column and the last row are never taps.  A cell outside
// [0, cols - 1) x [0, rows - 1) holds a value no depth is within 200 of, and what is counted is the taps WITHIN 200 of the centre:
//     count = (Dp == 0) ? n_window : n_window - n_near        (the reference counts  |Dp - tap| > 200 || Dp == 0  over the clipped window)
// -- integers, the same number.  The centre is read from the image itself (a pixel of the last column is a centre, never a tap).
#define KT_SD_RIM 3
#define KT_SD_LW (32 + 2 * KT_SD_RIM)
#define KT_SD_FAR 0x00ffffffu
struct kt_sd_args {
    const uint16_t* depth; float* scaled; kt_pixrec* rec; const uint8_t* colors; const float* nmap;
    int cols, rows; kt_intr intr; int angle_color, passes;
    float* tile_raw; int tl2, tcols;
    int strips_x;   // workgroups per strip row (the fused launch numbers them linearly)
};
__device__ __forceinline__ void kt_scale_depth_block(const kt_sd_args& a, int block_x, int block_y, int tx_, int ty_)
{
    const uint16_t* __restrict__ depth = a.depth; float* __restrict__ scaled = a.scaled; kt_pixrec* __restrict__ rec = a.rec;
    const uint8_t* __restrict__ colors = a.colors; const float* __restrict__ nmap = a.nmap;
    const int cols = a.cols, rows = a.rows, angle_color = a.angle_color, passes = a.passes, tl2 = a.tl2, tcols = a.tcols;
    const kt_intr intr = a.intr;
    float* __restrict__ tile_raw = a.tile_raw;
    __shared__ unsigned int s_win[(32 + 2 * KT_SD_RIM) * KT_SD_LW];   // up to 4 passes of 8 rows + the rim
    __shared__ unsigned int s_tile[4];   // raw maxima of the strip's tiles, as the bit patterns of non-negative floats (ordered like them)
    const int tid = ty_ * 32 + tx_;
    if (tile_raw && tid < 4) s_tile[tid] = 0u;
    const int bx = block_x * 32, by = block_y * passes * 8;
    if (angle_color) {
        const int lh = passes * 8 + 2 * KT_SD_RIM;
        for (int i = tid; i < lh * KT_SD_LW; i += 256) {
            const int ty = i / KT_SD_LW, tx = i - ty * KT_SD_LW;
            const int gx = bx + tx - KT_SD_RIM, gy = by + ty - KT_SD_RIM;
            const bool tap = gx >= 0 && gy >= 0 && gx < cols - 1 && gy < rows - 1;
            s_win[i] = tap ? (unsigned int)depth[gy * cols + gx] : KT_SD_FAR;
        }
    }
    __syncthreads();
    const int x = tx_ + bx;
    for (int pass = 0; pass < passes; ++pass) {
        const int y = ty_ + by + pass * 8;
        float out = 0.0f;
        if (x < cols && y < rows) {
            int Dp = depth[y * cols + x];
            float xl = ((float)x - intr.cx) / intr.fx;
            float yl = ((float)y - intr.cy) / intr.fy;
            float lambda = __builtin_sqrtf(__builtin_fmaf(xl, xl, yl * yl) + 1);
            if (angle_color) {
                const int ky = 7, kx = 7;
                const int ty = min(y - ky / 2 + ky, rows - 1), tx = min(x - kx / 2 + kx, cols - 1);
                const int n_window = max(ty - max(y - ky / 2, 0), 0) * max(tx - max(x - kx / 2, 0), 0);
                const unsigned int* w = &s_win[(ty_ + pass * 8) * KT_SD_LW + tx_];   // the window's top-left cell
                int n_near = 0;
#pragma unroll
                for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx) {
                        unsigned int d;
                        asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"((unsigned int)Dp), "v"(w[dy * KT_SD_LW + dx]));
                        n_near += d <= 200u ? 1 : 0;
                    }
                const int count = Dp == 0 ? n_window : n_window - n_near;
                out = (count > 5) ? (float)(-Dp) * lambda / 1000.f : (float)Dp * lambda / 1000.f;
            } else
                out = (float)Dp * lambda / 1000.f;
            scaled[y * cols + x] = out;
            if (rec) {
                float nx = nmap[y * cols + x];
                float nz = nmap[(y + 2 * rows) * cols + x];
                if (nz < 0) nz = -nz;
                kt_pixrec r;
                r.dp = out;
                r.wrkc = (angle_color ? fminf(1.0f, nz / KT_RGB_VIEW_ANGLE_WEIGHT) : 1.0f) * 2.0f;
                const uint8_t* c = &colors[3 * (y * cols + x)];
                r.rgbf = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | (kt_isnan(nx) ? KT_REC_NORMAL_NAN : 0u) |
                         ((angle_color && kt_isnan(nx)) ? KT_REC_STALE_NZ : 0u);
#if KT_REC_BYTES == 16
                r.pad = 0;
#endif
                rec[y * cols + x] = r;
            }
        }
        if (tile_raw) {
            // a wave is two rows of 32 pixels; a tile takes 2^tl2 consecutive lanes of each: fold them, then one LDS atomic per tile and row
            float m = fabsf(out);   // (0 outside the image)
            for (int off = 1; off < (1 << tl2) && off < 32; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            if ((tx_ & ((1 << tl2) - 1)) == 0) atomicMax(&s_tile[tx_ >> tl2], __float_as_uint(m));
        }
    }
    if (tile_raw) {
        __syncthreads();
        const int g = tx_, tx = (block_x * 32 >> tl2) + g;
        if (ty_ == 0 && g < (32 >> tl2) && tx < tcols) tile_raw[block_y * tcols + tx] = __uint_as_float(s_tile[g]);
    }
}

__global__ __launch_bounds__(256) void kt_scale_depth_kernel(const kt_sd_args a) { kt_scale_depth_block(a, blockIdx.x, blockIdx.y, threadIdx.x, threadIdx.y); }

// The tracker's frame preparation in one launch behind kt_pyramid01_kernel: pyramid levels 2 and 3 (from the level-1 depth) and scaleDepth + pixel
// records (from the level-0 normal map) depend on that launch only, not on each other -- one grid, the pyramid's 4 x 4 level-3 tiles first (the
// longer chain of barriers), then the strips: a launch boundary less on the read-ahead stream and the two latency chains side by side.
__global__ __launch_bounds__(256) void kt_prepare_fused_kernel(const kt_pyr_args pa, const kt_sd_args sa, int n23x, int n23)
{
    if ((int)blockIdx.x < n23) { kt_pyramid23_block<4>(pa, blockIdx.x % n23x, blockIdx.x / n23x, threadIdx.x); return; }
    const int b = blockIdx.x - n23;
    kt_scale_depth_block(sa, b % sa.strips_x, b / sa.strips_x, threadIdx.x & 31, threadIdx.x >> 5);
}

// Coarse map of the largest |scaled depth| per T x T pixel tile: lets the interval pre-pass bound, per voxel column, how far
// from the camera an update is still possible (sdf >= -trunc needs |v| <= Dp + trunc).  One wave per tile.  T = 8, 16 or 32: the
// smallest that keeps the map within KT_DPT_MAX_TILES entries (it is staged in LDS by the interval kernel): 8 at 640x480, 16 at 1280x960.
// The raw maxima come out of kt_scale_depth_kernel itself (round 5).
#define KT_DPT_MAX_TILES 8192
static int kt_dpt_log2(int cols, int rows)
{
    for (int l = 3; l <= 5; ++l)
        if (kt_div_up(cols, 1 << l) * kt_div_up(rows, 1 << l) <= KT_DPT_MAX_TILES && kt_div_up(cols, 32) * kt_div_up(rows, 32) <= 2048) return l;
    return 0;   // too many tiles even at 32 x 32: no depth-range prune
}
// What the interval pre-pass looks up is the DILATED map: entry (tx, ty) = the largest raw maximum of the 3 x 3 tiles around it, so that a
// piece of a column whose padded pixel bounding box is at most two tiles wide is bounded by the ONE entry under its centre.  Layout of the
// buffer (KT_DPT_FLOATS): dilated fine map [0 ..), dilated 32 x 32 pixel map [KT_DPT_MAX_TILES ..) for the coarse pass, the overall
// maximum at [KT_DPT_MAX_TILES + KT_DPT_COARSE_TILES], raw fine map (scratch of this stage) behind it.
#define KT_DPT_COARSE_TILES 2048
#define KT_DPT_RAW (KT_DPT_MAX_TILES + KT_DPT_COARSE_TILES + 1)
#define KT_DPT_FLOATS (KT_DPT_RAW + KT_DPT_MAX_TILES)
// ONE workgroup finishes the map from the raw fine maxima (round 4: a dilation launch and a coarse launch, 5 us of launch latency each):
// raw fine map into LDS; its 3 x 3 dilation out; raw 32 x 32 pixel maxima from the LDS copy, their dilation and the overall maximum out.
__global__ __launch_bounds__(1024) void kt_tile_finish_kernel(float* __restrict__ dpmax, int tl2, int tcols, int trows, int tc32, int tr32)
{
    __shared__ float raw[KT_DPT_MAX_TILES];
    __shared__ float raw32[KT_DPT_COARSE_TILES];
    __shared__ float wmax[16];
    const int n = tcols * trows;
    for (int i = threadIdx.x; i < n; i += 1024) raw[i] = dpmax[KT_DPT_RAW + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int tx = i % tcols, ty = i / tcols;
        float v = 0.0f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = tx + dx, y = ty + dy;
                if (x >= 0 && x < tcols && y >= 0 && y < trows) v = fmaxf(v, raw[y * tcols + x]);
            }
        dpmax[i] = v;
    }
    const int f = 5 - tl2;
    float all = 0.0f;
    for (int i = threadIdx.x; i < tc32 * tr32; i += 1024) {
        const int cx = i % tc32, cy = i / tc32;
        float v = 0.0f;
        for (int dy = 0; dy < (1 << f); ++dy)
            for (int dx = 0; dx < (1 << f); ++dx) {
                const int tx = (cx << f) + dx, ty = (cy << f) + dy;
                if (tx < tcols && ty < trows) v = fmaxf(v, raw[ty * tcols + tx]);
            }
        raw32[i] = v;
        all = fmaxf(all, v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tc32 * tr32; i += 1024) {
        const int cx = i % tc32, cy = i / tc32;
        float v = 0.0f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = cx + dx, y = cy + dy;
                if (x >= 0 && x < tc32 && y >= 0 && y < tr32) v = fmaxf(v, raw32[y * tc32 + x]);
            }
        dpmax[KT_DPT_MAX_TILES + i] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) all = fmaxf(all, __shfl_xor(all, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = all;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) all = fmaxf(all, wmax[w]);
        dpmax[KT_DPT_MAX_TILES + KT_DPT_COARSE_TILES] = all;
    }
}
static kt_sd_args kt_sd_args_fill(const uint16_t* depth_raw, float* depth_raw_scaled, kt_pixrec* rec, const uint8_t* colors, const float* nmap_curr, int cols, int rows,
                                  const kt_intr& intr, int angle_color, float* dpmax)
{
    kt_sd_args a;
    a.depth = depth_raw; a.scaled = depth_raw_scaled; a.rec = rec; a.colors = colors; a.nmap = nmap_curr;
    a.cols = cols; a.rows = rows; a.intr = intr; a.angle_color = angle_color;
    a.tl2 = (dpmax && rec) ? kt_dpt_log2(cols, rows) : 0;
    a.passes = a.tl2 ? (1 << a.tl2) / 8 : 1;
    a.tcols = a.tl2 ? kt_div_up(cols, 1 << a.tl2) : 0;
    a.tile_raw = a.tl2 ? dpmax + KT_DPT_RAW : nullptr;
    a.strips_x = kt_div_up(cols, 32);
    return a;
}
static void kt_launch_tile_finish(kt_ctx* c, const kt_sd_args& a, float* dpmax)
{
    if (a.tl2) hipLaunchKernelGGL(kt_tile_finish_kernel, dim3(1), dim3(1024), 0, c->stream, dpmax, a.tl2, a.tcols, kt_div_up(a.rows, 1 << a.tl2), kt_div_up(a.cols, 32), kt_div_up(a.rows, 32));
}
// scaleDepth + the pixel records and, when a depth-range map is wanted, its raw tile maxima from the same launch and the finishing launch
static void kt_launch_scale_depth(kt_ctx* c, const uint16_t* depth_raw, float* depth_raw_scaled, kt_pixrec* rec, const uint8_t* colors, const float* nmap_curr,
                                  int cols, int rows, const kt_intr& intr, int angle_color, float* dpmax)
{
    const kt_sd_args a = kt_sd_args_fill(depth_raw, depth_raw_scaled, rec, colors, nmap_curr, cols, rows, intr, angle_color, dpmax);
    hipLaunchKernelGGL(kt_scale_depth_kernel, dim3(a.strips_x, kt_div_up(rows, 8 * a.passes)), dim3(32, 8), 0, c->stream, a);
    kt_launch_tile_finish(c, a, dpmax);
}

struct kt_tsdf23_args {
    const kt_pixrec* rec;
    int16_t* volume;
    uchar4* color;
    const float* vgz;
    const float* zs;
    const unsigned int* tasks;     // compact list of (wave-column, z-chunk) units that contain work (kt_tsdf_tasks_kernel)
    const unsigned int* task_count;
    const unsigned int* wrange;    // per wave-column: union of its 64 column intervals, z0 | z1 << 16
    const float2* walk0;           // per column: (v_x, v_y) of the reference walk at z = the wave-column's first z
    const float* dpmax;            // [ceil(rows / T)][ceil(cols / T)] largest |scaled depth| per pixel tile (kt_scale_depth_kernel + kt_tile_finish_kernel), or null
    int dpt_log2;                  // log2 T (kt_dpt_log2)
    unsigned int* updated;  // optional counter (U of SURVEY 8d)
    kt_mat33 Ri;            // Rcurr_inv
    float tx, ty, tz;
    kt_intr intr;
    float cell_x, cell_y, cell_z;
    float tranc_dist;
    int wx, wy, wz;         // voxel wrap, normalised to [0, N)
    int cols, rows, N;
    const kt_frame_params* fp;  // when set: Ri / t come from the device (kt_frame_params) instead of the fields above
    unsigned char* bricks;      // optional [nb^3], nb = N / 32: set to 1 when a NEGATIVE tsdf is stored into the 32^3 storage brick
    int nb;                     // (raycast skips bricks that hold no negative value, see kt_raycast_kernel)
    // Planning ahead (kt_integrate_plan): the pre-pass runs on a PREDICTED pose, one frame early and off the frame's critical path.
    // The pose the frame ends up with may differ from the prediction by a rotation of at most theta and a translation of at most
    // tau; a voxel's camera coordinates then move by eps <= pm_A * p_z + pm_B (pm_A = 1.01 theta kappa, pm_B = 1.01 tau, kappa =
    // |p| / p_z at the image corner), and every bound of the pre-pass is widened by that much.  Both 0: the pose is the frame's own.
    float pm_A, pm_B;
    int wcl;                       // log2 of the wave-column's width in x: 5 = 32 x 2 columns, 4 = 16 x 4 (kt_tsdf_wcl)
};

// pose override shared by the two integrate kernels (wave-uniform scalar loads)
__device__ __forceinline__ bool kt_tsdf_pose_from_device(kt_tsdf23_args& a)
{
    if (!a.fp) return false;
#pragma unroll
    for (int k = 0; k < 9; ++k) a.Ri.m[k] = a.fp->Rinv[k];
    a.tx = a.fp->t[0]; a.ty = a.fp->t[1]; a.tz = a.fp->t[2];
    return a.fp->skip != 0;
}

// Conservative z-interval of one voxel column inside the (margin-padded) view frustum.  Only used to
// skip iterations whose in-image test must fail; every kept iteration runs the reference's exact test,
// so results do not depend on how tight this is.  p(z) = A + z*B in camera coordinates.
__device__ __forceinline__ void kt_clip_halfline(float alpha, float beta, float& lo, float& hi)
{
    // alpha + beta * z >= 0
    if (beta > 1e-12f) lo = fmaxf(lo, -alpha / beta);
    else if (beta < -1e-12f) hi = fminf(hi, -alpha / beta);
    else if (alpha < 0) { lo = 1e30f; hi = -1e30f; }
}

// Work decomposition of the voxel pass.  A unit of work ("task") is one wave-column -- 64 consecutive storage x of one y --
// times one chunk of KT_TSDF_ZCHUNK z indices.  Only ~5% of the N^2/64 x N/16 units intersect the view frustum, so:
//   1. kt_tsdf_interval_kernel   one thread per column: conservative z-interval inside the frustum; per wave-column the union;
//   2. kt_tsdf_tasks_kernel      one workgroup: prefix sums over the wave-columns -> compact task list, dealt by cost (no atomics);
//   3. kt_tsdf23_kernel          a fixed grid of KT_TSDF_WAVES waves strides over the list, so every wave that is launched
//                                has voxels to update and the SIMDs stay full of loads in flight.
// A task replays the incremental float walk of v_x / v_y from z = 0 to its first z (quirk A.17: the values are DEFINED by
// repeated +=; pure ALU, ~0.5 us), then runs the reference loop body 4 z-steps at a time with the loads of the 4 steps batched.
// (measured, 512^3 orbit / 768^3 static: ZCHUNK 8 -> -1% / -17%, 32 -> -4% / -2%; 4096 waves -> -4% / -10%, 16384 -> 0% / +2%)
#ifndef KT_TSDF_ZCHUNK
#define KT_TSDF_ZCHUNK 16
#endif
#ifndef KT_TSDF_UNROLL
#define KT_TSDF_UNROLL 4
#endif
#ifndef KT_TSDF_WPB
#define KT_TSDF_WPB 4         // waves per workgroup of the lean voxel kernel (its z table is per workgroup)
#endif
#ifndef KT_TSDF_WAVES
#define KT_TSDF_WAVES 8192
#endif
// Shape of a wave-column: 2^wcl consecutive storage x of 64 / 2^wcl consecutive y, a property of the launch (kt_tsdf23_args::wcl).
// 64 x 1 wastes a third of the lanes at the left / right frustum faces; 32 x 2 halves that and still moves 64 B of tsdf + 128 B of
// colour per row and access; 16 x 4 follows the frustum faces and depth discontinuities closer still (lane efficiency 0.59 -> 0.64 on
// the 512^3 orbit: launch 31.0 -> 29.0 us) but its 32-byte rows cost the dense 1280x960 @ 768^3 case 4.6 % (0.720 -> 0.753 ms).
// kt_tsdf_wcl picks per launch (rule and measurements: profiles/r03_experiments.md); KT_TSDF_WCX=16|32 in the environment overrides.
static size_t kt_tsdf_max_wave_cols(int N)   // wave-columns of an N^3 volume under either shape
{
    const size_t a = (size_t)kt_div_up(N, 32) * kt_div_up(N, 2), b = (size_t)kt_div_up(N, 16) * kt_div_up(N, 4);
    return a > b ? a : b;
}
static int kt_tsdf_wcl(int cols, int rows, int N)
{
    static const int forced = []() { const char* e = getenv("KT_TSDF_WCX"); const int v = e ? atoi(e) : 0; return v == 16 ? 4 : (v == 32 ? 5 : 0); }();
    if (forced) return forced;
    // pixels per voxel column: few -> the volume is sparse in the image (interval ends dominate) -> squarer wave-columns
    return (double)cols * rows <= 1.5 * (double)N * N ? 4 : 5;
}
// cache policy of the voxel kernel's volume accesses (buffer aux bits: 2 = nt).  Measured in round 3 (profiles/r03_experiments.md): nt
// loads +17 % launch time on the orbit (the words a frame updates were written by the frame before: nt gives up those hits), nt stores
// within noise on both workloads.
#ifndef KT_TSDF_LD_AUX
#define KT_TSDF_LD_AUX 0
#endif
#ifndef KT_TSDF_ST_AUX
#define KT_TSDF_ST_AUX 0
#endif
#define KT_TASK_HEAD 16   // words of a task list's head: [0] number of tasks, [1 + k] first task of XCD k (k = 0..8)

__device__ __forceinline__ int kt_wave_min(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int kt_wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// Pre-pass 1: the conservative z-interval of every voxel column, stored as (z0 | z1 << 16) per storage column (empty = N | 0),
// and per wave-column the union of its 64 intervals.  grid = (ceil(N / 64), ceil(N / 4)), 256 threads = 2 x 2 wave-columns.
__global__ __launch_bounds__(256) void kt_tsdf_interval_kernel(const kt_tsdf23_args a_in, unsigned int* __restrict__ wrange,
                                                               float2* __restrict__ walk0)
{
    kt_tsdf23_args a = a_in;
    const bool skip = kt_tsdf_pose_from_device(a);
    const int N = a.N;
    // a workgroup covers 2 x 2 wave-columns
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    const int xg = blockIdx.x * 2 + (wave_ & 1), yg = blockIdx.y * 2 + (wave_ >> 1);
    const int WX = 1 << a.wcl, WY = 64 >> a.wcl;
    const int sx = xg * WX + (lane_ & (WX - 1));
    const int sy = yg * WY + (lane_ >> a.wcl);
    const bool column = sx < N && sy < N;
    // dynamic LDS, sized to the two maps of this image (19 + 1.2 KB at 640x480): with the 40 KB of the largest case allocated
    // statically only three workgroups fit a CU and the 1024 workgroups of a 512^3 launch need two rounds
    extern __shared__ float s_maps[];
    float* const s_dpmax = s_maps;
    float* const s_dp32 = s_maps + ((((a_in.cols + (1 << a_in.dpt_log2) - 1) >> a_in.dpt_log2) * ((a_in.rows + (1 << a_in.dpt_log2) - 1) >> a_in.dpt_log2) + 3) & ~3);   // 32 x 32 pixel level
    const int tl2 = a.dpt_log2;
    const float Tinv = 1.0f / (float)(1 << tl2);
    const int tcols = (a.cols + (1 << tl2) - 1) >> tl2, trows = (a.rows + (1 << tl2) - 1) >> tl2;
    const bool prune = a.dpmax != nullptr && tl2 != 0;
    int z0 = N, z1 = 0;
    float v_g_x = 0.f, v_g_y = 0.f, ax = 0.f, ay = 0.f, az = 1.f, bx = 0.f, by = 0.f, bz = 0.f;
    float flo = 1e30f, fhi = -1e30f, nlo = 1e30f, nhi = -1e30f;
    const bool live = column && !skip;  // a frame parked for the host's shift path has no work
    if (live) {
        int x = sx - a.wx; if (x < 0) x += N;
        int y = sy - a.wy; if (y < 0) y += N;
        const float* Ri = a.Ri.m;
        v_g_x = __builtin_fmaf((float)x + 0.5f, a.cell_x, -a.tx);
        v_g_y = __builtin_fmaf((float)y + 0.5f, a.cell_y, -a.ty);
        const float v_g_z0 = __builtin_fmaf(0 + 0.5f, a.cell_z, -a.tz);
        // camera coordinates (unscaled) at z index 0 and the per-index step: p(z) = A + z * B
        ax = Ri[0] * v_g_x + Ri[1] * v_g_y + Ri[2] * v_g_z0;
        ay = Ri[3] * v_g_x + Ri[4] * v_g_y + Ri[5] * v_g_z0;
        az = Ri[6] * v_g_x + Ri[7] * v_g_y + Ri[8] * v_g_z0;
        bx = Ri[2] * a.cell_z; by = Ri[5] * a.cell_z; bz = Ri[8] * a.cell_z;
        const float m = 2.0f;       // pixel margin: this linear model and the kernel's accumulated walk differ by << 1 pixel
        const float znear = 0.05f;  // below this depth the pixel bounds are not trusted
        const float lo = 0.0f, hi = (float)(N - 1);
        flo = lo; fhi = hi;         // frustum part: p_z >= znear and the four image sides padded by m pixels
        // (planned ahead: the frame's own camera coordinates p satisfy |p - p^| <= eps(p^) = pm_A p^_z + pm_B, so an image side
        // f p_x - u p_z >= 0 is implied by f p^_x - u p^_z + (f + |u|) eps >= 0: the same half-plane with u moved by (f + |u|) pm_A
        // and the constant (f + |u|) pm_B added)
        kt_clip_halfline(az - znear + 1.1f * a.pm_B + a.pm_A * znear, bz, flo, fhi);
        float ul = -0.5f - m - a.intr.cx, uh = (float)a.cols - 0.5f + m - a.intr.cx;
        float vl = -0.5f - m - a.intr.cy, vh = (float)a.rows - 0.5f + m - a.intr.cy;
        const float gul = a.intr.fx + fabsf(ul), guh = a.intr.fx + fabsf(uh), gvl = a.intr.fy + fabsf(vl), gvh = a.intr.fy + fabsf(vh);
        ul -= gul * a.pm_A; uh += guh * a.pm_A; vl -= gvl * a.pm_A; vh += gvh * a.pm_A;
        kt_clip_halfline(a.intr.fx * ax - ul * az + gul * a.pm_B, a.intr.fx * bx - ul * bz, flo, fhi);
        kt_clip_halfline(uh * az - a.intr.fx * ax + guh * a.pm_B, uh * bz - a.intr.fx * bx, flo, fhi);
        kt_clip_halfline(a.intr.fy * ay - vl * az + gvl * a.pm_B, a.intr.fy * by - vl * bz, flo, fhi);
        kt_clip_halfline(vh * az - a.intr.fy * ay + gvh * a.pm_B, vh * bz - a.intr.fy * by, flo, fhi);
        // near slab: -cell <= p_z <= znear.  There the pixel coordinates are ill-conditioned, so the slab is kept without
        // testing them -- but a voxel that close to the camera plane can only project into the image when it is also within
        // |p_x|, |p_y| <= znear * (image half-size / f) ~ 0.06 m of the optical axis; columns that stay 0.2 m away skip it.
        nlo = lo; nhi = hi;
        kt_clip_halfline(az + a.cell_z + a.cell_x + a.cell_y + 1.1f * a.pm_B, bz, nlo, nhi);
        kt_clip_halfline(znear + 1.1f * a.pm_B + a.pm_A * znear - az, -bz, nlo, nhi);
        if (nlo <= nhi) {
            const float pxa = ax + nlo * bx, pxb = ax + nhi * bx, pya = ay + nlo * by, pyb = ay + nhi * by;
            const float rlim = 0.2f + 2.0f * a.pm_B;
            const bool far_x = (pxa > rlim && pxb > rlim) || (pxa < -rlim && pxb < -rlim);
            const bool far_y = (pya > rlim && pyb > rlim) || (pya < -rlim && pyb < -rlim);
            if (far_x || far_y) { nlo = 1e30f; nhi = -1e30f; }
        }
        // (the depth-range prune of the frustum part follows below, once the workgroup has staged the tile maps)
    }
    // the two levels of the map are staged in LDS only by workgroups that have a column to prune (most lie outside the frustum)
    const bool pruned = live && prune && flo <= fhi && !(nlo <= nhi);
    float Dall = 0.0f;   // the largest entry of the map: the bound of a piece that covers too many tiles to look at
    if (prune && __syncthreads_or(pruned)) {
        const int nc = ((a.cols + 31) >> 5) * ((a.rows + 31) >> 5);
        for (int i = threadIdx.x; i < tcols * trows; i += 256) s_dpmax[i] = a.dpmax[i];
        for (int i = threadIdx.x; i < nc; i += 256) s_dp32[i] = a.dpmax[KT_DPT_MAX_TILES + i];
        Dall = a.dpmax[KT_DPT_MAX_TILES + KT_DPT_COARSE_TILES];
        __syncthreads();
    }
    if (live) {
        // Depth-range prune of the frustum part.  A voxel farther from the camera than D + trunc, D = the scaled depth of the pixel it
        // projects to, has sdf < -trunc, and one whose pixel has no depth is never updated.  The voxels of the column project onto a
        // straight image segment; it is cut into pieces of equal length in z, and for each piece D is bounded by the tile maxima under the
        // bounding box of its two end pixels (padded by 2.5 pixels: rounding to the pixel, and the difference between this linear model and
        // the kernel's accumulated walk).  Inside a piece |v|^2 = v_g_x^2 + v_g_y^2 + ((z + 0.5) cell_z - t_z)^2 <= (D + trunc)^2 bounds
        // z from above.  The pieces are visited from the far end, first in steps of about one 32-pixel tile against the coarse map, then
        // inside the first coarse piece that keeps anything in steps of one fine tile; the first fine piece that keeps anything ends the
        // interval (everything in front of a surface is updated, so the near end stays the frustum's).  Skipped when the near slab is
        // kept: the segment's end is ill-defined there.  (One bound for the whole column from 32 x 32 tiles, round 1: 7.4 M lane z-steps on
        // the 512^3 orbit; this: 6.9 M; an exact per-wave-column interval would need 5.5 M.)
        if (pruned) {
            const float r2xy = v_g_x * v_g_x + v_g_y * v_g_y;
            const float inv_cell = 1.0f / a.cell_z;
            const int tc32 = (a.cols + 31) >> 5, tr32 = (a.rows + 31) >> 5;
            // pixel of the column at (fractional) z; the reciprocal is the hardware's (1 ulp): far inside the 2.5-pixel pad
            auto pix = [&](float z, float& u, float& v) {
                const float inv = __builtin_amdgcn_rcpf(az + z * bz);
                u = a.intr.fx * (ax + z * bx) * inv + a.intr.cx;
                v = a.intr.fy * (ay + z * by) * inv + a.intr.cy;
            };
            // Bound of |scaled depth| under the padded bounding box of two pixels from the dilated map of tile size 2^l2: a box at most two
            // tiles wide lies inside the 3 x 3 tiles around the tile of its centre, whose maximum is that entry.  (Wider: the map's maximum.)
            // (planned ahead: the frame's own pixel lies within (f + |u - c|) eps / (p_z - eps) of the predicted one; the pad of a piece
            // is taken at its near end, where that is largest)
            auto bound = [&](float u0, float v0, float u1, float v1, float pad, const float* map, int l2, int mc, int mr) -> float {
                const float ti = 1.0f / (float)(1 << l2);
                if (fmaxf(fabsf(u1 - u0), fabsf(v1 - v0)) + 2.0f * pad > (float)(2 << l2)) return Dall;
                const int tx = min(mc - 1, max(0, (int)floorf(0.5f * (u0 + u1) * ti)));
                const int ty = min(mr - 1, max(0, (int)floorf(0.5f * (v0 + v1) * ti)));
                return map[ty * mc + tx];
            };
            // the part of [z0, z1] that can still be updated when no pixel under it is deeper than D: false = none, else its upper end.
            // |v(z)|^2 = r2xy + w(z)^2 with w(z) = (z + 0.5) cell_z - t_z linear in z, so the smallest |v| of the piece needs no square root;
            // two voxels of slack in R cover the rounding of z to voxel centres and of this arithmetic.
            auto keep = [&](float z0, float z1, float D, float& top) -> bool {
                const float R = (D + a.tranc_dist) * 1.001f + 1e-4f + 2.0f * a.cell_z + a.pm_B;   // (the camera centre moves by <= tau)
                const float w0 = __builtin_fmaf(z0 + 0.5f, a.cell_z, -a.tz), w1 = __builtin_fmaf(z1 + 0.5f, a.cell_z, -a.tz);
                const float wmin2 = (w0 <= 0.0f && w1 >= 0.0f) ? 0.0f : fminf(w0 * w0, w1 * w1);
                const float s2 = R * R - r2xy;
                if (!(wmin2 <= s2)) return false;
                top = fminf(z1, (a.tz + __builtin_amdgcn_sqrtf(s2) * 1.00001f) * inv_cell - 0.5f);
                return true;
            };
            // 2.5 pixels of rounding + the planning margin at (fractional) z, where the column projects to (u, v): a camera-space
            // displacement of eps moves the pixel by at most (f + |u - c|) eps / (p_z - eps)
            auto pad_at = [&](float z, float u, float v) -> float {
                if (a.pm_A == 0.0f && a.pm_B == 0.0f) return 2.5f;
                const float pz = az + z * bz, eps = a.pm_A * pz + a.pm_B;
                const float g = fmaxf(a.intr.fx, a.intr.fy) + fmaxf(fabsf(u - a.intr.cx), fabsf(v - a.intr.cy)) + 3.0f;
                return 2.5f + g * eps * __builtin_amdgcn_rcpf(fmaxf(pz - eps, 1e-3f)) * 1.01f;
            };
            float ua, va, ub, vb;
            pix(flo, ua, va);
            pix(fhi, ub, vb);
            const float len = fmaxf(fabsf(ub - ua), fabsf(vb - va));
            // Coarse pass: pieces of about one 32-pixel tile, from the far end; a piece that keeps nothing under the coarse (larger) bound
            // keeps nothing under the fine one.  The refinement of a surviving piece (a fine pass of up to 8 pieces) is kept OUT of the coarse
            // loop: lanes reach their first surviving piece at different trip counts, and a fine pass nested in the loop would run once
            // per trip for whichever lanes need it then (measured: that divergence was half of the kernel).  So, in rounds: every lane scans
            // to its next surviving coarse piece, then all of them refine together; a lane whose refinement keeps nothing scans on.  After
            // two refinements a lane takes the coarse answer.
            const int Kc = min(32, (int)(len * (1.0f / 32.0f)) + 1);
            const float dzc = (fhi - flo) / (float)Kc;
            float zc1 = fhi, uc1 = ub, vc1 = vb, hi_new = -1e30f;
            bool any = false;
            int ic = Kc - 1;
            for (int round = 0; round < 3 && ic >= 0 && !any; ++round) {
                float zc0 = flo, uc0 = ua, vc0 = va, top = 0.0f;
                bool found = false;
                for (; ic >= 0; --ic) {   // (a) scan
                    zc0 = ic == 0 ? flo : flo + dzc * (float)ic;
                    uc0 = ua; vc0 = va;
                    if (ic != 0) pix(zc0, uc0, vc0);
                    if (keep(zc0, zc1, bound(uc0, vc0, uc1, vc1, fmaxf(pad_at(zc0, uc0, vc0), pad_at(zc1, uc1, vc1)), s_dp32, 5, tc32, tr32), top)) { found = true; break; }
                    zc1 = zc0; uc1 = uc0; vc1 = vc0;
                }
                if (!found) break;
                if (tl2 == 5 || round == 2) { hi_new = top; any = true; break; }
                // (b) refine [zc0, zc1]
                const float lenf = fmaxf(fabsf(uc1 - uc0), fabsf(vc1 - vc0));
                const int Kf = min(8, (int)(lenf * Tinv) + 1);
                const float dzf = (zc1 - zc0) / (float)Kf;
                float zf1 = zc1, uf1 = uc1, vf1 = vc1;
                for (int jf = Kf - 1; jf >= 0; --jf) {
                    const float zf0 = jf == 0 ? zc0 : zc0 + dzf * (float)jf;
                    float uf0 = uc0, vf0 = vc0;
                    if (jf != 0) pix(zf0, uf0, vf0);
                    // (Ending the column exactly -- walking the surviving piece against a full-resolution dilated depth map -- was measured in
                    // round 3: lane efficiency 0.591 -> 0.620, the voxel kernel 2.5 us shorter, this pre-pass 13 -> 33 us; planned ahead
                    // it runs next to the odometry chain, which it slows by 10 us: 5 % fewer frames per second.  Not kept.)
                    if (keep(zf0, zf1, bound(uf0, vf0, uf1, vf1, fmaxf(pad_at(zf0, uf0, vf0), pad_at(zf1, uf1, vf1)), s_dpmax, tl2, tcols, trows), top)) { hi_new = top; any = true; break; }
                    zf1 = zf0; uf1 = uf0; vf1 = vf0;
                }
                zc1 = zc0; uc1 = uc0; vc1 = vc0;   // nothing kept in there: scan on below it
                --ic;
            }
            if (!any) { flo = 1e30f; fhi = -1e30f; }
            else fhi = fminf(fhi, hi_new);
        }
        float l = 1e30f, h = -1e30f;
        if (flo <= fhi) { l = fminf(l, flo); h = fmaxf(h, fhi); }
        if (nlo <= nhi) { l = fminf(l, nlo); h = fmaxf(h, nhi); }
        if (l <= h) {
            z0 = max(0, (int)floorf(l) - 2);
            z1 = min(N, (int)ceilf(h) + 3);
            if (z0 >= z1) { z0 = N; z1 = 0; }
        }
    }
    const int wz0 = kt_wave_min(z0), wz1 = kt_wave_max(z1);
    const int XG = (N + WX - 1) / WX, YG = (N + WY - 1) / WY;
    if (lane_ == 0 && xg < XG && yg < YG) wrange[(size_t)yg * XG + xg] = (unsigned int)wz0 | ((unsigned int)wz1 << 16);
    // Checkpoint of the incremental walk (quirk A.17: v_x, v_y are DEFINED by repeated float +=) at the wave-column's first z:
    // walked once per column here (wave-uniform trip count), so a voxel task only replays from there to its own chunk.
    if (walk0 && wz0 < wz1 && column)   // (a plan made ahead of its frame leaves them to the frame's set-up kernel: they need the frame's own pose)
        walk0[(size_t)sy * N + sx] = kt_tsdf_walk_checkpoint(a.Ri.m, a.tx, a.ty, a.tz, a.cell_x, a.cell_y, a.cell_z, a.intr.fx, a.intr.fy, sx, sy, a.wx, a.wy, N, wz0);
}

// Pre-pass 2: compact task list, dealt by cost.  One workgroup; thread t owns a contiguous run of wave-columns; task = yg | xg << 16 |
// chunk << 24 (wave-column (xg, yg), kt_tsdf_interval_kernel).  In list order the tasks of two x-neighbouring wave-columns alternate
// chunk by chunk: the 4 waves of a workgroup then work on (a, c), (b, c), (a, c + 1), (b, c + 1), i.e. on both 64-byte halves of the
// same 128-byte tsdf lines at the same time.  Head of the list (KT_TASK_HEAD words): [0] number of tasks, [1 + k] first task of XCD k.
// Until round 3 the list was used in that order, cut into eighths by count.  But the voxel kernel of the 512^3 orbit is ONE task per resident wave, and the eight tasks that
// share a SIMD differ in cost: the SIMDs of a launch finish +-20 % apart (profiles/r02_tsdf23_whatif.md) and the launch takes as long
// as the slowest.  A task's cost is, to first order, its number of 4-step batches b (1..4), known here.  So: (1) XCD k takes the k-th
// eighth of the list by COST, not by count (still a contiguous band of image rows for its L2); (2) inside an XCD the tasks are sorted
// by b, most expensive first, stably -- position p goes to workgroup (p / 4) % 256, wave p % 4, and dispatch order puts workgroup j on
// CU j % 32 (a performance assumption, nothing else): every SIMD then gets one task of each cost octile; (3) odd rows of 128 are
// reversed (snake), so that the SIMD with the most expensive task of one round gets the cheapest of the next.  scripts/lane_model.py
// (dealing section) models max / mean of the per-SIMD sums: 1.28 list order -> 1.09-1.22; measured (r03 call 10, same box): the
// orbit launch alone 34.0 -> 31.3 us, the 768^3 launch 0.725 -> 0.699 ms, every parity test unchanged (the order of the tasks cannot
// change a voxel: each is written by exactly one task).  The sort is stable, so neighbours in list order stay neighbours inside a class.
// Counting sort without atomics and without memory traffic per task: a thread's whole run of wave-columns goes to ONE XCD (the one its
// exclusive cost prefix falls into -- 1/1024 of the list is fine as a granule), so a thread needs four class counters, in registers; one
// block scan of {cost, 4 class counts} gives every thread its offset inside each class of its XCD.  Two walks over the thread's
// wave-columns (count, place).  Measured around it in round 3 (profiles/r03_experiments.md): 32 counters per thread in LDS touched per
// task (27 us against 20 for this one and 8 for the unsorted list); the counts taken in the interval kernel + a one-workgroup scan + one
// wave per pair placing its tasks by ballot ranks (three launches, 41 us of plan-stream work in all): both slower for the frame, because
// whatever runs on the plan stream sits on CUs next to the odometry chain, whose 256 one-per-CU workgroups then run a second round.
__global__ __launch_bounds__(1024) void kt_tsdf_tasks_kernel(const unsigned int* __restrict__ wrange, int M, int XG,
                                                             unsigned int* __restrict__ tasks, unsigned int* __restrict__ task_count)
{
    constexpr int NB = KT_TSDF_ZCHUNK / KT_TSDF_UNROLL;          // batches of a full task
    static_assert(NB == 4, "class layout assumes 4 batches per task");
    __shared__ unsigned int wave_tot[5][16];
    __shared__ unsigned char xs[1024];           // XCD of every thread's run
    __shared__ unsigned int cbase[4][9];         // exclusive class-count prefix at the first thread of XCD x (x = 8: totals)
    __shared__ unsigned int bbase[33];           // first list position of bucket 4 x + class
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (((M + 1023) / 1024) + 1) & ~1;               // even: runs start on an even wave-column
    const int i0 = min(M, tid * per), i1 = min(M, i0 + per);
    // the thread's tasks in list order: fn(key, batches)
    auto walk = [&](auto&& fn) {
        for (int i = i0; i < i1; i += 2) {
            int c0[2] = {1, 1}, c1[2] = {0, 0}, z0[2] = {0, 0}, z1[2] = {0, 0};
            unsigned int key[2] = {0, 0};
            for (int k = 0; k < 2; ++k) {
                if (i + k >= i1) continue;
                const unsigned int r = wrange[i + k];
                z0[k] = (int)(r & 0xffffu); z1[k] = (int)(r >> 16);
                if (z0[k] < z1[k]) { c0[k] = z0[k] / KT_TSDF_ZCHUNK; c1[k] = (z1[k] - 1) / KT_TSDF_ZCHUNK; }
                key[k] = (unsigned int)((i + k) / XG) | ((unsigned int)((i + k) % XG) << 16);
            }
            const int lo = min(c0[0] <= c1[0] ? c0[0] : INT_MAX, c0[1] <= c1[1] ? c0[1] : INT_MAX);
            const int hi = max(c0[0] <= c1[0] ? c1[0] : -1, c0[1] <= c1[1] ? c1[1] : -1);
            for (int c = lo; c <= hi; ++c)
                for (int k = 0; k < 2; ++k)
                    if (c >= c0[k] && c <= c1[k]) {
                        const int za = max(z0[k], c * KT_TSDF_ZCHUNK), zb = min(z1[k], (c + 1) * KT_TSDF_ZCHUNK);
                        fn(key[k] | ((unsigned int)c << 24), (zb - za + KT_TSDF_UNROLL - 1) / KT_TSDF_UNROLL);
                    }
        }
    };
    // walk 1: cost and class counts (class q = NB - batches: 0 = most expensive)
    unsigned int mine[5] = {0, 0, 0, 0, 0};
    walk([&](unsigned int, int b) {
        mine[0] += (unsigned int)b;
        mine[1] += b == 4 ? 1u : 0u; mine[2] += b == 3 ? 1u : 0u; mine[3] += b == 2 ? 1u : 0u; mine[4] += b == 1 ? 1u : 0u;
    });
    unsigned int incl[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) incl[v] = mine[v];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const unsigned int up = __shfl_up(incl[v], off, 64);
            if (lane >= off) incl[v] += up;
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int v = 0; v < 5; ++v) wave_tot[v][wave] = incl[v];
    }
    __syncthreads();
    unsigned int excl[5], total[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        unsigned int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned int t = wave_tot[v][w];
            if (w < wave) base += t;
            tot += t;
        }
        excl[v] = base + incl[v] - mine[v];
        total[v] = tot;
    }
    const unsigned int per_xcd_w = max(1u, (total[0] + 7u) / 8u);
    const int my_x = (int)min(7u, excl[0] / per_xcd_w);
    xs[tid] = (unsigned char)my_x;
    __syncthreads();
    {
        const int prev_x = tid == 0 ? -1 : (int)xs[tid - 1];
        for (int x = prev_x + 1; x <= my_x; ++x) {     // (XCDs no run falls into get an empty part)
#pragma unroll
            for (int q = 0; q < 4; ++q) cbase[q][x] = excl[1 + q];
        }
        if (tid == 1023)
            for (int x = my_x + 1; x <= 8; ++x) {
#pragma unroll
                for (int q = 0; q < 4; ++q) cbase[q][x] = total[1 + q];
            }
    }
    __syncthreads();
    if (tid < 64) {   // bucket sizes in (XCD, class) order -> exclusive scan over 32 lanes
        const int x = (tid & 31) >> 2, q = tid & 3;
        const unsigned int size = tid < 32 ? cbase[q][x + 1] - cbase[q][x] : 0u;
        unsigned int in = size;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const unsigned int up = __shfl_up(in, off, 64);
            if (lane >= off) in += up;
        }
        if (tid < 32) bbase[tid] = in - size;
        if (tid == 31) { bbase[32] = in; task_count[0] = in; task_count[9] = in; }
        if (tid < 32 && q == 0) task_count[1 + x] = in - size;
    }
    __syncthreads();
    // walk 2: place.  Position inside the XCD's part = the bucket's start + the thread's offset in its class + a running count.
    const unsigned int start = bbase[4 * my_x], n = bbase[4 * my_x + 4] - start;
    unsigned int off0 = bbase[4 * my_x + 0] - start + (excl[1] - cbase[0][my_x]);
    unsigned int off1 = bbase[4 * my_x + 1] - start + (excl[2] - cbase[1][my_x]);
    unsigned int off2 = bbase[4 * my_x + 2] - start + (excl[3] - cbase[2][my_x]);
    unsigned int off3 = bbase[4 * my_x + 3] - start + (excl[4] - cbase[3][my_x]);
    walk([&](unsigned int key, int b) {
        unsigned int p = b == 4 ? off0 : (b == 3 ? off1 : (b == 2 ? off2 : off3));
        off0 += b == 4 ? 1u : 0u; off1 += b == 3 ? 1u : 0u; off2 += b == 2 ? 1u : 0u; off3 += b == 1 ? 1u : 0u;
        const unsigned int row = p >> 7;
        if ((row & 1u) && (row + 1u) * 128u <= n) p = row * 128u + (127u - (p & 127u));
        tasks[start + p] = key;
    });
}

// Round 4: the same list without a serial walk per thread (kt_tsdf_tasks_kernel above: two walks over a thread's run of wave-columns, and
// the active wave-columns are a contiguous band, so 15 % of the threads walked all the tasks: 20-22 us in ONE workgroup).  Two launches:
//   kt_tsdf_tasks_scan_kernel (one workgroup)
//     (1) the class counts {cost, n4, n3, n2, n1} of a PAIR of x-neighbouring wave-columns follow from its two z-ranges in O(1) -- the
//         interior chunks of a column are full (4 batches), only its first and last chunk can be shorter -- so counting needs no loop;
//     (2) their exclusive prefixes over all pairs (list order) go to memory behind the list's head: any wave can then place any pair;
//     (3) the XCD of a pair is where its cost prefix falls (granule = one pair; the serial form's granule was a thread's run);
//   kt_tsdf_tasks_place_kernel (one WAVE per pair, the whole GPU)
//     (4) one lane per (chunk, column) slot of the pair in list order, ranks inside a class from ballots.
// Same order rules as above: XCD parts by cost, inside a part stable by class (most batches first), odd rows of 128 reversed.
// Layout of the list's head buffer (KT_TASK_HEAD_WORDS(pairs) words): [0, 16) the head proper, [16, 52) cbase[4][9], [52, 85) bbase[33],
// [85] share of an XCD, [96 + v (P + 1) + i] exclusive prefix of quantity v over pairs.
#define KT_TASK_CBASE 16
#define KT_TASK_BBASE 52
#define KT_TASK_SHARE 85
#define KT_TASK_PREF 96
static size_t kt_task_head_words(int N) { return KT_TASK_PREF + 5 * ((kt_tsdf_max_wave_cols(N) + 1) / 2 + 1); }
__device__ __forceinline__ void kt_task_counts(unsigned int r, unsigned int (&cnt)[5])
{
    const int z0 = (int)(r & 0xffffu), z1 = (int)(r >> 16);
    if (z0 >= z1) return;
    const int c0 = z0 / KT_TSDF_ZCHUNK, c1 = (z1 - 1) / KT_TSDF_ZCHUNK;
    const int bf = (min(z1, (c0 + 1) * KT_TSDF_ZCHUNK) - z0 + KT_TSDF_UNROLL - 1) / KT_TSDF_UNROLL;   // first chunk (the only one if c0 == c1)
    cnt[0] += (unsigned int)bf; cnt[5 - bf] += 1u;
    if (c1 > c0) {
        const int bl = (z1 - c1 * KT_TSDF_ZCHUNK + KT_TSDF_UNROLL - 1) / KT_TSDF_UNROLL;
        cnt[0] += (unsigned int)bl + 4u * (unsigned int)(c1 - c0 - 1); cnt[5 - bl] += 1u; cnt[1] += (unsigned int)(c1 - c0 - 1);
    }
}
__device__ __forceinline__ void kt_pair_counts(const unsigned int* __restrict__ wrange, int M, int i, unsigned int (&cnt)[5])
{
#pragma unroll
    for (int v = 0; v < 5; ++v) cnt[v] = 0;
    kt_task_counts(wrange[2 * i], cnt);
    if (2 * i + 1 < M) kt_task_counts(wrange[2 * i + 1], cnt);
}
__global__ __launch_bounds__(1024) void kt_tsdf_tasks_scan_kernel(const unsigned int* __restrict__ wrange, int M, unsigned int* __restrict__ head)
{
    static_assert(KT_TSDF_ZCHUNK / KT_TSDF_UNROLL == 4, "class layout assumes 4 batches per task");
    __shared__ unsigned int wave_tot[5][16];
    __shared__ unsigned int cbase[4][9];         // class-count prefix at the first pair of XCD x (x = 8: totals)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = (M + 1) >> 1;
    const int S = P + 1;
    unsigned int* const pref = head + KT_TASK_PREF;
    // contiguous pairs per thread, thread sums, block scan
    const int pp = (P + 1023) / 1024;
    const int i0 = min(P, tid * pp), i1 = min(P, i0 + pp);
    unsigned int mine[5] = {0, 0, 0, 0, 0};
    for (int i = i0; i < i1; ++i) {
        unsigned int cnt[5];
        kt_pair_counts(wrange, M, i, cnt);
#pragma unroll
        for (int v = 0; v < 5; ++v) mine[v] += cnt[v];
    }
    unsigned int incl[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) incl[v] = mine[v];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const unsigned int up = __shfl_up(incl[v], off, 64);
            if (lane >= off) incl[v] += up;
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int v = 0; v < 5; ++v) wave_tot[v][wave] = incl[v];
    }
    __syncthreads();
    unsigned int run[5], total[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        unsigned int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned int t = wave_tot[v][w];
            if (w < wave) base += t;
            tot += t;
        }
        run[v] = base + incl[v] - mine[v];
        total[v] = tot;
    }
    const unsigned int per_xcd = max(1u, (total[0] + 7u) / 8u);
    // per-pair prefixes out; XCD parts: pair i belongs to XCD min(7, cost prefix / share), and the first pair of every XCD leaves the
    // class prefixes there (cbase).  The XCD of the pair in front of a thread's run follows from that pair's own counts.
    int x_prev = -1;
    if (i0 > 0 && i0 < P) {
        unsigned int cnt[5];
        kt_pair_counts(wrange, M, i0 - 1, cnt);
        x_prev = (int)min(7u, (run[0] - cnt[0]) / per_xcd);
    }
    for (int i = i0; i < i1; ++i) {
        const int x_here = (int)min(7u, run[0] / per_xcd);
        for (int x = x_prev + 1; x <= x_here; ++x) {   // (XCDs no pair falls into get an empty part)
#pragma unroll
            for (int q = 0; q < 4; ++q) cbase[q][x] = run[1 + q];
        }
        x_prev = x_here;
        unsigned int cnt[5];
        kt_pair_counts(wrange, M, i, cnt);
#pragma unroll
        for (int v = 0; v < 5; ++v) { pref[v * S + i] = run[v]; run[v] += cnt[v]; }
    }
    if (i1 == P && i0 < P) {   // the thread that owns the last pair closes the table
        for (int x = x_prev + 1; x <= 8; ++x) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cbase[q][x] = total[1 + q];
        }
#pragma unroll
        for (int v = 0; v < 5; ++v) pref[v * S + P] = total[v];
        head[KT_TASK_SHARE] = per_xcd;
    }
    __syncthreads();
    if (tid < 64) {   // bucket sizes in (XCD, class) order -> exclusive scan over 32 lanes
        const int x = (tid & 31) >> 2, q = tid & 3;
        const unsigned int size = tid < 32 ? cbase[q][x + 1] - cbase[q][x] : 0u;
        unsigned int in = size;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const unsigned int up = __shfl_up(in, off, 64);
            if (lane >= off) in += up;
        }
        if (tid < 32) head[KT_TASK_BBASE + tid] = in - size;
        if (tid == 31) { head[KT_TASK_BBASE + 32] = in; head[0] = in; head[9] = in; }
        if (tid < 32 && q == 0) head[1 + x] = in - size;
    }
    if (tid < 36) head[KT_TASK_CBASE + tid] = cbase[tid / 9][tid % 9];
}
__global__ __launch_bounds__(256) void kt_tsdf_tasks_place_kernel(const unsigned int* __restrict__ wrange, int M, int XG, const unsigned int* __restrict__ head,
                                                                   unsigned int* __restrict__ tasks)
{
    const int lane = threadIdx.x & 63;
    const int P = (M + 1) >> 1, S = P + 1;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= P) return;
    const unsigned int* const pref = head + KT_TASK_PREF;
    const unsigned int pc = pref[pair];
    if (pref[pair + 1] == pc) return;   // no task in this pair (wave-uniform)
    const int x = (int)min(7u, pc / head[KT_TASK_SHARE]);
    const unsigned int start = head[KT_TASK_BBASE + 4 * x], n = head[KT_TASK_BBASE + 4 * x + 4] - start;
    unsigned int running[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) running[q] = head[KT_TASK_BBASE + 4 * x + q] - start + (pref[(1 + q) * S + pair] - head[KT_TASK_CBASE + q * 9 + x]);
    const int k = lane & 1, wc = 2 * pair + k;
    const unsigned int r = wc < M ? wrange[wc] : (unsigned int)KT_TSDF_ZCHUNK;   // (an odd M leaves the last pair half empty: z0 > z1)
    const int z0 = (int)(r & 0xffffu), z1 = (int)(r >> 16);
    const bool some = z0 < z1;
    const int c0 = some ? z0 / KT_TSDF_ZCHUNK : INT_MAX, c1 = some ? (z1 - 1) / KT_TSDF_ZCHUNK : -1;
    const int lo = kt_wave_min(c0), hi = kt_wave_max(c1);
    const unsigned int key = (unsigned int)(wc / XG) | ((unsigned int)(wc % XG) << 16);
    for (int cb = lo; cb <= hi; cb += 32) {
        const int c = cb + (lane >> 1);
        const bool valid = c >= c0 && c <= c1;
        const int za = max(z0, c * KT_TSDF_ZCHUNK), zb = min(z1, (c + 1) * KT_TSDF_ZCHUNK);
        const int b = valid ? (zb - za + KT_TSDF_UNROLL - 1) / KT_TSDF_UNROLL : 0;
        unsigned int p = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(b == 4 - q);
            if (b == 4 - q) p = running[q] + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            running[q] += (unsigned int)__builtin_popcountll(m);
        }
        if (valid) {
            const unsigned int row = p >> 7;
            if ((row & 1u) && (row + 1u) * 128u <= n) p = row * 128u + (127u - (p & 127u));
            tasks[start + p] = key | ((unsigned int)c << 24);
        }
    }
}

// One in-flight batch of KT_TSDF_UNROLL consecutive z steps of a wave: everything phase A produces for phase B.
struct kt_tsdf_batch {
    bool in_img[KT_TSDF_UNROLL];
    unsigned int off[KT_TSDF_UNROLL];   // storage element index of the voxel (pointer path)
    int sz[KT_TSDF_UNROLL];             // storage z (wave-uniform; buffer path)
    int bz[KT_TSDF_UNROLL];             // (storage z >> 5) * nb * nb (wave-uniform)
    float vgz[KT_TSDF_UNROLL];
    kt_pixrec rec[KT_TSDF_UNROLL];
    short tsdf_raw[KT_TSDF_UNROLL];
    unsigned int col[KT_TSDF_UNROLL];   // the uchar4 {r, g, b, weight} as one word
};

// Buffer path (BUF, every N < 1024): the two volumes are addressed through buffer descriptors (descriptors for the pixel records
// and the brick flags as well cost more SGPRs than the kernel has: 50 spills, more VALU than before).
// A voxel access is then `buffer_load v, v_column_offset, s[descriptor], s_plane_offset`: the z-plane's byte offset lives in an SGPR
// (z is wave-uniform), the lane's offset inside the plane is loop-invariant, and no per-step VALU address arithmetic is left
// (the pointer path spends 2-5 VALU per access on 64-bit adds, a quarter-rate multiply for the pixel index included).
// Needs 32-bit byte offsets: N^3 * 4 < 2^32.
struct kt_tsdf_bufs { __amdgpu_buffer_rsrc_t vol, col; };

template <bool BUF>
__device__ __forceinline__ void kt_tsdf_load_voxel(const kt_tsdf23_args& a, const kt_tsdf_bufs& m, kt_tsdf_batch& b, int u, unsigned int col_base, unsigned int plane)
{
    if constexpr (BUF) {
        const unsigned int zoff = (unsigned int)b.sz[u] * plane;   // elements; wave-uniform
        b.tsdf_raw[u] = __builtin_amdgcn_raw_buffer_load_b16(m.vol, col_base * 2u, zoff * 2u, KT_TSDF_LD_AUX);
        b.col[u] = __builtin_amdgcn_raw_buffer_load_b32(m.col, col_base * 4u, zoff * 4u, KT_TSDF_LD_AUX);
    } else {
        b.tsdf_raw[u] = a.volume[b.off[u]];
        b.col[u] = *(const unsigned int*)&a.color[b.off[u]];
    }
}

// 1 / d, correctly rounded, without the scaling wrapper of the IEEE expansion (v_div_scale x2, v_div_fixup): the refinement chain
// hipcc emits for 1.0f / d, which is exact as it stands whenever nothing in it can overflow or go denormal.  The caller guarantees
// 2^-20 <= |d| <= 2^20 (checked once per task at the ends of the z chunk: d is monotone in z); kt_debug_exact_ops compares it
// with the division on every float of that range.
__device__ __forceinline__ float kt_rcp_exact(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r, q);
}

// The voxel kernel runs the reference loop body KT_TSDF_UNROLL z-steps at a time in two phases:
//   issue    projection of the 4 voxels, then ALL their loads at once -- the 16-byte pixel record and, speculatively for every
//            voxel that projects into the image, its tsdf and colour words (the update predicate needs the record, so waiting
//            for it first would double the exposed latency);
//   consume  sdf test, running-average update, stores (only of words that changed).
// Every lane of the wave walks the same z sequence: each volume access is one contiguous 128 B / 256 B segment.
// The column intervals only select the tasks: inside a task every lane runs the reference's in-image test on every step (a voxel
// outside its conservative interval fails it by construction), which is cheaper than two more compares per step.
// (Non-temporal volume loads / stores, meant to keep the streamed volume from evicting the pixel records in L2, measured 17% slower
// on the 512^3 orbit and neutral on the dense 768^3 case.)
// Variants kept switchable for A/B runs (scripts/exp_variants.sh; measured in profiles/r03_tsdf23_variants_call*.log):
#ifndef KT_TSDF_DEFER
#define KT_TSDF_DEFER 1   // 1: the volume words are loaded only for voxels that a cheap bound on the update predicate lets through
#endif                    //    (0: speculatively for every voxel that projects into the image; -5 % time, -9 % fetched bytes)
#ifndef KT_TSDF_MARK
#define KT_TSDF_MARK 1    // 1: the running average's division as a table reciprocal + one correction step (0: the IEEE sequence; -0.5 %)
#endif
// (packed fp32 for the projection -- depths and reciprocal chains of two z-steps per v_pk_fma_f32, both image coordinates of a step
// in one -- was measured with them: 1077 -> 1056 static VALU, launch time unchanged to 0.1 us on both workloads.  On gfx950 a
// v_pk_fma_f32 occupies the issue port 1.5x as long as a v_fma_f32 (profiles/r03_valu_rates.md), and hipcc spends the rest on moves.)

template <bool COUNT, bool BUF, bool FAST>
__device__ __forceinline__ void kt_tsdf_issue(const kt_tsdf23_args& a, const kt_tsdf_bufs& m, kt_tsdf_batch& b, int zb, int z_end, bool lane_ok,
                                              unsigned int col_base, unsigned int plane, float v_z, float& v_x, float& v_y, float dvx,
                                              float dvy, float tab_vgz, float tab_zs, int tab_base, float r8)
{
    const int N = a.N;
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u) {
        const int z = zb + u;
        const int zz = min(z, N - 1);  // wave-uniform
        // wave-uniform table entries: broadcast from the lane that holds them (no memory access in the loop)
        const int tl = zz - tab_base;
        b.vgz[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tab_vgz), tl));
        const float z_scaled = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tab_zs), tl));
        const float d = __builtin_fmaf(r8, z_scaled, v_z);
        const float inv_z = FAST ? kt_rcp_exact(d) : 1.0f / d;
        const int coo_x = kt_f2i_rn(__builtin_fmaf(v_x, inv_z, a.intr.cx));
        const int coo_y = kt_f2i_rn(__builtin_fmaf(v_y, inv_z, a.intr.cy));
        // 0 <= coo < size as ONE unsigned compare per axis
        // z_end (wave-uniform): the last batch of a task may reach past the chunk, whose next z belong to another task
        b.in_img[u] = lane_ok && z < z_end && !(inv_z < 0) && (unsigned int)coo_x < (unsigned int)a.cols && (unsigned int)coo_y < (unsigned int)a.rows;
        int sz = zz + a.wz; if (sz >= N) sz -= N;
        b.bz[u] = (sz >> KT_BRICK_LOG2) * a.nb * a.nb;
        // coo_x, coo_y < 2^24 inside the image: the 24-bit multiply is exact there (and full rate, unlike v_mul_lo_u32)
        const unsigned int pix = b.in_img[u] ? kt_mad24((unsigned int)coo_y, (unsigned int)a.cols, (unsigned int)coo_x) : 0u;
        // (the gather through a buffer descriptor -- offset by one more 24-bit multiply instead of the 64-bit v_mad_u64_u32 of this address
        // -- was measured in round 3: launch time unchanged, 14 more spilled SGPRs)
        b.rec[u] = a.rec[pix];
        if constexpr (BUF) b.sz[u] = sz;
        else b.off[u] = col_base + (unsigned int)sz * plane;
        v_x += dvx;  // the walk advances on every step, also on skipped ones
        v_y += dvy;
    }
#if !KT_TSDF_DEFER
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u)
        if (b.in_img[u]) kt_tsdf_load_voxel<BUF>(a, m, b, u, col_base, plane);
#endif
}

template <bool COUNT, bool BUF>
__device__ __forceinline__ void kt_tsdf_consume(const kt_tsdf23_args& a, const kt_tsdf_bufs& m, kt_tsdf_batch& b, float v_g_part_norm,
                                                float tranc_dist_inv, unsigned int& n_upd, int brick_xy, unsigned int& n_img,
                                                unsigned int col_base, unsigned int plane, const float* __restrict__ s_rcp)
{
#if KT_TSDF_DEFER
    // The update predicate (dp != 0 and sdf >= -trunc, sdf = |dp| - |v|) needs only the pixel record: |v|^2 <= (|dp| + trunc)^2 is
    // necessary for it (1e-5 of relative slack covers the roundings of both sides: five float operations of <= 2^-24 each), so the
    // tsdf and colour words are requested only for voxels that pass this bound -- on the 512^3 orbit a quarter of the voxels that
    // project into the image lie behind the surface and never needed their 6 bytes.
    bool may[KT_TSDF_UNROLL];
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u) {
        const float s = fabsf(b.rec[u].dp) + a.tranc_dist;
        may[u] = b.in_img[u] && b.rec[u].dp != 0 && __builtin_fmaf(b.vgz[u], b.vgz[u], v_g_part_norm) <= s * s * 1.00001f;
    }
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u)
        if (may[u]) kt_tsdf_load_voxel<BUF>(a, m, b, u, col_base, plane);
    // (Splitting the record -- 4 bytes of scaled depth first, {weight, rgb} only for these voxels -- was measured too: 83 % / 75 % of
    // the voxels in the image pass the bound on the two workloads, so nearly every record is fetched anyway and the second array
    // only adds lines: +8 % fetched bytes, +7 % / +12 % time.)
#endif
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u) {
        if (COUNT && b.in_img[u]) ++n_img;   // diagnostics: voxel steps that project into the image
#if KT_TSDF_DEFER
        if (!may[u]) continue;
#else
        if (!b.in_img[u]) continue;
#endif
        const float dp = b.rec[u].dp;
        const float Dp_scaled = fabsf(dp);          // a negative scaled depth flags "no colour" (tsdf_volume.cu:520-527, :590-594)
        const bool no_color = dp < 0.0f;
        // Classification with v_sqrt_f32 (<= 1 ulp): t = sdf / trunc to within 2e-5.  t > 1.001: free space, tsdf = min(1, t) is
        // exactly 1 whatever the last bit of the square root is.  t < -1.001: sdf < -trunc, no update.  Only voxels in between (the
        // truncation band and a hair around its two edges) take the correctly rounded sqrt, behind a wave-uniform branch.
        const float r2 = __builtin_fmaf(b.vgz[u], b.vgz[u], v_g_part_norm);
        float sdf = Dp_scaled - __builtin_amdgcn_sqrtf(r2);
        const float t = sdf * tranc_dist_inv;
        const bool is_free = t > 1.001f;
        const bool band = !is_free && t >= -1.001f;   // NaN (never: r2 >= 0) would fall out of both
        if (__builtin_amdgcn_ballot_w64(band) != 0) {
            asm volatile("; exact sqrt" ::: "memory");  // a real branch: if-converted, the 16-instruction correctly rounded sqrt runs for every voxel
            if (band) sdf = Dp_scaled - __builtin_sqrtf(r2);
        }
        if (!(dp != 0 && (is_free || (band && sdf >= -a.tranc_dist)))) continue;
        if (COUNT) ++n_upd;
        const unsigned int c = b.col[u];
        const float weight_prev = (float)(c >> 24);
        // free voxel that already holds F = 1 (raw 32767): (1 * W + 1) / (W + 1) == 1 exactly, the stored value stays
        const bool touch = !(is_free && b.tsdf_raw[u] == KT_DIVISOR);
        if (__builtin_amdgcn_ballot_w64(touch) != 0) {
            if (touch) {
                const float tsdf = is_free ? 1.0f : fminf(1.0f, sdf * tranc_dist_inv);
                const float tsdf_prev = kt_unpack_tsdf(b.tsdf_raw[u]);
#if KT_TSDF_MARK
                // (F W + tsdf) / (W + 1), correctly rounded without the IEEE division sequence: y = RN(1 / (W + 1)) from a 256-entry
                // table in LDS (built with the division at kernel start), q = RN(n y), one exact residual, one correction (Markstein:
                // with a correctly rounded y the corrected q is the correctly rounded quotient).  kt_debug_div_check compares it with
                // the division for every finite float numerator and every divisor 1..256.
                const float num = __builtin_fmaf(tsdf_prev, weight_prev, tsdf), den = weight_prev + 1.0f, y = s_rcp[c >> 24];
                const float q0 = num * y;
                const short packed = kt_pack_tsdf(__builtin_fmaf(__builtin_fmaf(-den, q0, num), y, q0));
#else
                const short packed = kt_pack_tsdf(__builtin_fmaf(tsdf_prev, weight_prev, tsdf) / (weight_prev + 1.0f));
#endif
                if (packed != b.tsdf_raw[u]) {   // an unchanged word is not written back
                    if constexpr (BUF) __builtin_amdgcn_raw_buffer_store_b16(packed, m.vol, col_base * 2u, (unsigned int)b.sz[u] * plane * 2u, KT_TSDF_ST_AUX);
                    else a.volume[b.off[u]] = packed;
                    if (a.bricks && packed < 0) a.bricks[b.bz[u] + brick_xy] = 1;  // idempotent byte store, no atomics (a negative value
                }                                                                  // that stays was flagged when it was first stored)
            }
        }
        // weight: min(W + 1, 128) in the top byte, on the whole word (saturating add: W = 255 must not wrap)
        unsigned int o = min(__builtin_elementwise_add_sat(c, 0x01000000u), (c & 0x00ffffffu) | 0x80000000u);   // also for foreign weights above 128
        const unsigned int rgbf = b.rec[u].rgbf;
        // colour update iff (normal valid and not flagged "no colour") or the stored colour is (0, 0, 0)  (tsdf_volume.cu:623).
        // A voxel whose stored colour already equals the pixel's keeps it: with c == rgb the blend is
        // rint(RN(c (W + Wrkc(1 + e1))(1 + e2) / ((W + Wrkc)(1 + e3)))) = c for every weight (|error| <= 255 * 4 * 2^-24 << 0.5), and
        // the degenerate 0 / 0 case (W = Wrkc = 0, or a NaN weight) can only meet c = 0 = rgb, where the reference stores 0 too.
        const bool same_colour = ((c ^ rgbf) & 0xffffffu) == 0;
        const bool blend = ((!(rgbf & KT_REC_NORMAL_NAN) && !no_color) || (c & 0xffffffu) == 0) && !same_colour;
        if (__builtin_amdgcn_ballot_w64(blend) != 0) {
            if (blend) {
                // c' = clamp(rint(RN(n / den))) per channel: only the INTEGER is stored.  k = rint(n * v_rcp_f32(den)) is a candidate; it
                // is the reference's value whenever |n - k * den| < 0.4999 * den: the residual comes out of one FMA (relative error
                // 2^-24), so the true quotient is then within 0.49991 of k and its float rounding (|error| <= 2^-16 below 256) still
                // rounds to k.  n / den is a weighted mean of values in [0, 255], so k needs no clamp.  Otherwise -- a tie to within
                // 1e-4, 0 / 0, a NaN weight -- the IEEE divisions run (wave-uniform branch, rare).
                const float Wrkc = b.rec[u].wrkc;
                const float den = weight_prev + Wrkc;
                // numerators c_prev * W + Wrkc * c_new: the SECOND product is the fused one (oracle/_ref pins it at a .5 tie of the quotient)
                const float nx = __builtin_fmaf(Wrkc, (float)(rgbf & 0xffu), (float)(c & 0xffu) * weight_prev);
                const float ny = __builtin_fmaf(Wrkc, (float)((rgbf >> 8) & 0xffu), (float)((c >> 8) & 0xffu) * weight_prev);
                const float nz = __builtin_fmaf(Wrkc, (float)((rgbf >> 16) & 0xffu), (float)((c >> 16) & 0xffu) * weight_prev);
                const float rden = __builtin_amdgcn_rcpf(den);
                float kx = __builtin_rintf(nx * rden), ky = __builtin_rintf(ny * rden), kz = __builtin_rintf(nz * rden);
                const float lim = 0.4999f * den;
                const bool safe = fabsf(__builtin_fmaf(-kx, den, nx)) < lim && fabsf(__builtin_fmaf(-ky, den, ny)) < lim &&
                                  fabsf(__builtin_fmaf(-kz, den, nz)) < lim;   // false for NaN
                if (__builtin_amdgcn_ballot_w64(!safe) != 0) {
                    if (!safe) {
                        kx = (float)min(255, max(0, kt_f2i_rn(nx / den)));
                        ky = (float)min(255, max(0, kt_f2i_rn(ny / den)));
                        kz = (float)min(255, max(0, kt_f2i_rn(nz / den)));
                    }
                }
                o = (o & 0xff000000u) | (unsigned int)kx | ((unsigned int)ky << 8) | ((unsigned int)kz << 16);
            }
        }
        if (o != c) {   // a saturated free-space voxel in front of an unchanged pixel costs reads only
            if constexpr (BUF) __builtin_amdgcn_raw_buffer_store_b32(o, m.col, col_base * 4u, (unsigned int)b.sz[u] * plane * 4u, KT_TSDF_ST_AUX);
            else *(unsigned int*)&a.color[b.off[u]] = o;
        }
    }
}


#ifndef KT_TSDF_OCC
#define KT_TSDF_OCC 8   // waves per SIMD the register budget is cut for
#endif
#ifndef KT_TSDF_LEAN_DEFAULT
#define KT_TSDF_LEAN_DEFAULT 1   // 1: kt_tsdf23_lean_kernel (round 4) is the voxel kernel of every N < 1024 launch unless KT_TSDF_LEAN=0
#endif
template <bool COUNT, bool BUF>
__global__ __launch_bounds__(256, KT_TSDF_OCC) void kt_tsdf23_kernel(const kt_tsdf23_args a_in)
{
    kt_tsdf23_args a = a_in;
    kt_tsdf_bufs m = {};
#if KT_TSDF_MARK
    __shared__ float s_rcp_tab[256];
    s_rcp_tab[threadIdx.x] = 1.0f / (float)(threadIdx.x + 1);   // RN(1 / (W + 1)) for every weight byte W
    __syncthreads();
    const float* s_rcp = s_rcp_tab;
#else
    const float* s_rcp = nullptr;
#endif
    if constexpr (BUF) {   // descriptors from kernel arguments only: they stay in SGPRs
        const unsigned int nvox = (unsigned int)a_in.N * (unsigned int)a_in.N * (unsigned int)a_in.N;
        m.vol = __builtin_amdgcn_make_buffer_rsrc((void*)a_in.volume, 0, nvox * 2u, 0x00020000);
        m.col = __builtin_amdgcn_make_buffer_rsrc((void*)a_in.color, 0, nvox * 4u, 0x00020000);
    }
    // a parked frame does nothing here: its in-stream pre-pass left an empty task list, but a plan made ahead of the frame did not
    if (kt_tsdf_pose_from_device(a)) return;
    const int N = a.N;
    const int WX = 1 << a.wcl, WY = 64 >> a.wcl;
    const int lane = threadIdx.x & 63;
    const float* Ri = a.Ri.m;
    const float v_g_z0 = __builtin_fmaf(0 + 0.5f, a.cell_z, -a.tz);
    const float dvx = Ri[2] * a.cell_z * a.intr.fx;   // Rcurr_inv_0_z_scaled
    const float dvy = Ri[5] * a.cell_z * a.intr.fy;   // Rcurr_inv_1_z_scaled
    const float tranc_dist_inv = 1.0f / a.tranc_dist;
    const unsigned int plane = (unsigned int)N * (unsigned int)N;
    float r8 = Ri[8];
    asm volatile("" : "+v"(r8));   // keep it in a VGPR: fma(r8, z_scaled, v_z) then takes the broadcast z_scaled straight from its SGPR
    unsigned int n_upd = 0, n_batches = 0, n_tasks_done = 0, n_img = 0;
    // XCD-aware task order: workgroup b runs on XCD b % 8 (and each XCD has its own L2), so XCD k takes the k-th contiguous part of
    // the list (equal shares of the cost, kt_tsdf_tasks_kernel) -- a band of y, i.e. a band of image rows whose 12-byte pixel records
    // then stay in that one L2 -- and inside an XCD four consecutive tasks go to the 4 waves of one workgroup.
    const unsigned int t_begin = a.task_count[1 + (blockIdx.x & 7u)], t_end = a.task_count[2 + (blockIdx.x & 7u)];
    for (unsigned int t = t_begin + (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6); t < t_end; t += (gridDim.x >> 3) * 4u) {
        const unsigned int task = __builtin_amdgcn_readfirstlane(a.tasks[t]);
        const int yg = (int)(task & 0xffffu), xg = (int)((task >> 16) & 0xffu), chunk = (int)(task >> 24);
        const int sx = xg * WX + (lane & (WX - 1));
        const int sy = min(yg * WY + (lane >> a.wcl), N - 1);   // a row past the volume (odd N) only repeats the last one, never live
        const bool lane_ok = sx < N && yg * WY + (lane >> a.wcl) < N;
        // the wave-column's union of column intervals, cut to this chunk (wave-uniform)
        const unsigned int wr = __builtin_amdgcn_readfirstlane(a.wrange[(size_t)yg * ((N + WX - 1) / WX) + xg]);
        const int zc = (int)(wr & 0xffffu);
        const int wz0 = max(zc, chunk * KT_TSDF_ZCHUNK), wz1 = min((int)(wr >> 16), (chunk + 1) * KT_TSDF_ZCHUNK);
        if (wz0 >= wz1) continue;
        int x = sx - a.wx; if (x < 0) x += N;
        int y = sy - a.wy; if (y < 0) y += N;
        const float v_g_x = __builtin_fmaf((float)x + 0.5f, a.cell_x, -a.tx);
        const float v_g_y = __builtin_fmaf((float)y + 0.5f, a.cell_y, -a.ty);
        const float v_g_part_norm = __builtin_fmaf(v_g_x, v_g_x, v_g_y * v_g_y);
        const float v_z = __builtin_fmaf(Ri[8], v_g_z0, __builtin_fmaf(Ri[6], v_g_x, Ri[7] * v_g_y));
        // the reference's walk (tsdf_volume.cu:566-571, 634-640): resume from the wave-column's checkpoint (the value at its first z,
        // kt_tsdf_interval_kernel) and advance to this task's first z
        const float2 cp = a.walk0[(size_t)sy * N + min(sx, N - 1)];
        float v_x = cp.x, v_y = cp.y;
        {
            int z = zc;
            for (; z + 16 <= wz0; z += 16) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { v_x += dvx; v_y += dvy; }
            }
            for (; z < wz0; ++z) { v_x += dvx; v_y += dvy; }
        }
        const unsigned int col_base = (unsigned int)min(sx, N - 1) + (unsigned int)sy * (unsigned int)N;
        const int brick_xy = (sy >> KT_BRICK_LOG2) * a.nb + (min(sx, N - 1) >> KT_BRICK_LOG2);
        // the chunk's slice of the z-walk tables, one entry per lane (KT_TSDF_ZCHUNK + KT_TSDF_UNROLL <= 64 == wave size)
        const int tab_base = chunk * KT_TSDF_ZCHUNK;
        const float tab_vgz = a.vgz[min(tab_base + lane, N - 1)];
        const float tab_zs = a.zs[min(tab_base + lane, N - 1)];
        // The camera-frame depth d(z) = fma(R8, z_scaled(z), v_z) of a column is monotone in z, so its values inside the chunk lie between
        // those at the chunk's ends: when both are in [2^-20, 2^20] in magnitude and of one sign, every reciprocal of the task can use
        // the unwrapped refinement chain (kt_rcp_exact); a column that crosses the camera plane inside the chunk takes the division.
        const float zs_a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tab_zs), 0));
        const float zs_b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tab_zs), min(KT_TSDF_ZCHUNK + KT_TSDF_UNROLL - 1, N - 1 - tab_base)));
        const float d_a = __builtin_fmaf(r8, zs_a, v_z), d_b = __builtin_fmaf(r8, zs_b, v_z);
        const bool d_ok = fminf(fabsf(d_a), fabsf(d_b)) >= 0x1p-20f && fmaxf(fabsf(d_a), fabsf(d_b)) <= 0x1p20f && (d_a < 0) == (d_b < 0);
        const bool fast = __builtin_amdgcn_ballot_w64(!d_ok) == 0;
        // Latency is hidden by occupancy (8 tasks per SIMD) rather than by cross-iteration software pipelining: hipcc's s_waitcnt
        // insertion cannot count loads that are still in flight across a loop back-edge and falls back to vmcnt(0).  (A straight-line
        // 4-batch version with two batches in flight needs 96 VGPRs = 5 waves per SIMD and measured 15% / 40% slower at 512^3 / 768^3.)
        for (int zb = wz0; zb < wz1; zb += KT_TSDF_UNROLL) {
            kt_tsdf_batch cur;
            if (fast) kt_tsdf_issue<COUNT, BUF, true>(a, m, cur, zb, wz1, lane_ok, col_base, plane, v_z, v_x, v_y, dvx, dvy, tab_vgz, tab_zs, tab_base, r8);
            else kt_tsdf_issue<COUNT, BUF, false>(a, m, cur, zb, wz1, lane_ok, col_base, plane, v_z, v_x, v_y, dvx, dvy, tab_vgz, tab_zs, tab_base, r8);
            kt_tsdf_consume<COUNT, BUF>(a, m, cur, v_g_part_norm, tranc_dist_inv, n_upd, brick_xy, n_img, col_base, plane, s_rcp);
            if (COUNT) ++n_batches;
        }
        if (COUNT) ++n_tasks_done;
    }
    if (COUNT) {
        // wave-level sum, one atomic per wave and counter; [1] = wave batches, [2] = tasks with work (diagnostics)
        for (int off = 32; off > 0; off >>= 1) { n_upd += __shfl_down(n_upd, off, 64); n_img += __shfl_down(n_img, off, 64); }
        if (lane == 0) {
            if (n_upd) atomicAdd(a.updated, n_upd);
            if (n_batches) atomicAdd(a.updated + 1, n_batches);
            if (n_tasks_done) atomicAdd(a.updated + 2, n_tasks_done);
            if (n_img) atomicAdd(a.updated + 3, n_img);
        }
    }
}


// ---- round 4: the voxel kernel with its scalar half cut (KT_TSDF_LEAN=0 in the environment selects the kernel above) -----------------
// Round 3's counters: 100 VALU + 65 SALU + 4 VMEM wave-instructions per wave z-step, 38 spilled SGPRs.  The scalar half was bookkeeping:
// exec-mask regions around every predicate of the consume phase, the wave-uniform z arithmetic (table index, storage wrap, plane
// offsets x 2 and x 4, brick row) repeated per step, the kernel's 60 argument SGPRs, v_readlane broadcasts of the z tables.  Here:
//   * the per-z quantities {v_g_z, z_scaled, element offset of the storage plane, brick plane} of ALL z are built once per workgroup in
//     LDS (16 bytes per z) and fetched with ONE uniform ds_read_b128 per step -- the LDS pipe was idle, the scalar unit and the
//     v_readlane / s_nop pairs were not;
//   * the voxel's buffer offset is a VGPR add (no SGPR plane offset, no s_mul / s_lshl per access);
//   * the update predicate is evaluated ONCE, before the volume words are requested (round 3 evaluated a cheap necessary bound first
//     and the predicate again after the loads): tsdf = min(1, sdf / trunc) is exactly 1 on the free-space side whatever the last bit
//     of the square root, so the same guard band as before decides who takes the correctly rounded root, and the loads go out only
//     for voxels that WILL be updated;
//   * predicates are combined with non-short-circuit logic, the pixel index is computed unconditionally and selected;
//   * the kernel takes a cut-down argument block (kt_tsdf_lean_args: what the loop needs, nothing of the pre-pass).
// Stored bits are those of the kernel above (same expressions, same guard bands); tests/test_gpu_volume.py runs both.
struct kt_tsdf_ztab { float vgz, zs; unsigned int zoff2; int bz; };   // per z: walk tables, byte offset of the storage plane in the tsdf volume, brick plane index
struct kt_tsdf_lean_args {
    const kt_pixrec* rec;
    int16_t* volume;
    uchar4* color;
    const float* vgz;
    const float* zs;
    const unsigned int* tasks;
    const unsigned int* task_count;
    const unsigned int* wrange;
    const float2* walk0;
    unsigned int* updated;
    const kt_frame_params* fp;
    unsigned char* bricks;
    kt_mat33 Ri;
    float tx, ty, tz;
    kt_intr intr;
    float cell_x, cell_y, cell_z;
    float tranc_dist;
    int wx, wy, wz;
    int cols, rows, N;
    int nb, wcl;
};

// Issue cost on gfx950 (profiles/r04_valu_rates.md, 8 waves per SIMD, measured clock): v_fma / v_mul / v_add / v_sub_f32, v_add_u32,
// v_and_b32, v_mov_b32 take ~2.5 cycles per wave-instruction; EVERYTHING else -- compares, v_cndmask, min / max, shifts, every
// conversion, v_rndne, 24-bit multiplies, v_mad_u64_u32 -- ~4.2, v_rcp / v_sqrt 8.2.  The lean kernel is VALU-issue bound, so its inner loop
// trades the second kind for the first wherever the value is provably the same:
//   * round-to-nearest-even to an integer = add 1.5 * 2^23 (one v_add_f32; exact for |x| < 2^22) and read the integer off the mantissa,
//     instead of v_rndne_f32 + v_cvt_i32_f32;
//   * byte offsets come from the z table and are ADDED to a loop-invariant column offset; x 2 for the colour volume is an add too;
//   * byte -> float by v_cvt_f32_ubyteN + v_mul_f32 instead of the 24-bit integer multiply + conversion hipcc selects.
#define KT_RNE_MAGIC 12582912.0f       // 1.5 * 2^23: ulp = 1, so x + MAGIC rounds x to the nearest integer, ties to even
#define KT_RNE_MAGIC_BITS 0x4B400000u
template <int N>
__device__ __forceinline__ float kt_ubyte_f32(unsigned int v)   // (float)((v >> 8 N) & 0xff) as ONE conversion (and nothing the compiler can fold into an integer multiply)
{
    float r;
    if constexpr (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(v));
    else if constexpr (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(v));
    else if constexpr (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(v));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ unsigned int kt_twice(unsigned int x)   // 2 x as v_add_u32 (hipcc would select the shift, which issues at half the rate)
{
    unsigned int r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float kt_min1(float x)   // min(1, x) as one v_min_f32 (fminf costs a canonicalising v_max_f32 on top; x is never NaN here)
{
    float r;
    asm("v_min_f32 %0, 1.0, %1" : "=v"(r) : "v"(x));
    return r;
}

// TOL: the second, explicitly named arithmetic contract of the voxel kernel ("survey-8c": SURVEY.md 8(c)'s parity policy and north_star's
// 1e-4 on floats instead of every stored bit; VERDICT r4 item 2 iii).  What it drops: the correctly rounded square root inside the
// truncation band (v_sqrt_f32, <= 1 ulp, everywhere), the Markstein correction of the running average (q = n * RN(1 / d): <= 1 ulp),
// the residual test of the colour blend (k = rint(n * v_rcp_f32(den)): wrong only next to a .5 tie).  What it keeps: the projection's
// exact reciprocal (the PIXEL a voxel reads must not change: "bit-exact on voxel indices"), every predicate, weight and store rule.
// Consequences, counted by tests/test_gpu_tol.py against the bit-exact kernel at BASELINE configs 2 / 3 / 5: tsdf shorts differ by at
// most 1 on a small fraction of the touched voxels, colour bytes by at most 1 at near-ties, weights on 0-2 voxels of 135 M (not a guarantee: the
// update predicate `sdf >= -trunc` is decided by v_sqrt_f32 here, so a voxel within the root's last bit of the -trunc edge can be updated by one
// kernel and not by the other -- advisor, round 5).  The bit-exact kernel stays
// the default of the library, of every parity test and of bench.py's headline.
// CT = 2, "speed of light" (round 6): a MEASUREMENT variant, never a product path.  Same task list, same loads, stores and
// predicates as the kernels above; the per-voxel arithmetic cut down to what the reference's own build flags execute (CMakeLists.txt:47:
// --prec-div=false --prec-sqrt=false, i.e. rcp.approx / sqrt.approx and no correction steps): v_rcp_f32 in the projection (no refinement
// chain, no range test per task), v_sqrt_f32 everywhere, the running average and the colour blend without their residual corrections, the
// stored tsdf unpacked with one multiply.  Its launch time on the same workload is what "the contract costs" means in numbers: if IT does not
// reach BASELINE.json's 0.60 of the HBM roofline, no arithmetic contract of this decomposition does (bench.py: roofline*.speed_of_light).
template <bool COUNT, bool FAST, bool NT, int CT = 0>
__device__ __forceinline__ void kt_tsdf_batch_lean(const kt_tsdf_lean_args& a, const kt_tsdf_bufs& m, const kt_tsdf_ztab* __restrict__ s_tab,
                                                   const float* __restrict__ s_rcp, int zb, int rem, unsigned int col_base2, int brick_xy,
                                                   float v_z, float& v_x, float& v_y, float dvx, float dvy, float r8, float v_g_part_norm,
                                                   float tranc_dist_inv, unsigned int& n_upd, unsigned int& n_img)
{
    constexpr bool TOL = CT >= 1, SOL = CT == 2;
    kt_pixrec rec[KT_TSDF_UNROLL];
    bool in_img[KT_TSDF_UNROLL];
    float r2[KT_TSDF_UNROLL];
    unsigned int toff[KT_TSDF_UNROLL];   // byte offset of the voxel in the tsdf volume (the colour volume: twice that)
    // loop constants the fast VALU operations read live in VGPRs: an SGPR operand halves the issue rate of v_fma / v_add_f32
    // (profiles/r04_valu_rates.md: 4.2 cycles against 2.4)
    float2 cxy = make_float2(a.intr.cx, a.intr.cy);
    float magic = KT_RNE_MAGIC;
    asm volatile("" : "+v"(cxy.x), "+v"(cxy.y), "+v"(magic), "+v"(r8));
    // ---- phase A: project the 4 voxels, gather their pixel records
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u) {
        const kt_tsdf_ztab e = s_tab[zb + u];   // wave-uniform address: one broadcast ds_read_b128
        const float d = __builtin_fmaf(r8, e.zs, v_z);
        const float inv_z = SOL ? __builtin_amdgcn_rcpf(d) : (FAST ? kt_rcp_exact(d) : 1.0f / d);
        const float px = __builtin_fmaf(v_x, inv_z, cxy.x), py = __builtin_fmaf(v_y, inv_z, cxy.y);
        unsigned int coo_x, coo_y;
        if constexpr (FAST) {
            // __float2int_rn off the mantissa: for |p| < 2^22 the sum is exactly MAGIC + rne(p); a larger |p|, an infinity or a NaN (none of
            // which the FAST path can produce: |1 / d| <= 2^20 and the walk is finite) leaves a word that fails the unsigned in-image test
            coo_x = __float_as_uint(px + magic) - KT_RNE_MAGIC_BITS;
            coo_y = __float_as_uint(py + magic) - KT_RNE_MAGIC_BITS;
        } else {
            coo_x = (unsigned int)kt_f2i_rn(px);
            coo_y = (unsigned int)kt_f2i_rn(py);
        }
        // the last batch of a task may reach past the chunk (whose next z belong to another task): those steps see an image of 0 rows
        const unsigned int rows_u = u < rem ? (unsigned int)a.rows : 0u;   // wave-uniform
        bool in = (coo_x < (unsigned int)a.cols) & (coo_y < rows_u);   // 0 <= coo < size as ONE unsigned compare per axis
        if constexpr (!FAST) in = in & !(inv_z < 0);   // FAST: the sign of d is constant over the task and part of the task's lane mask
        in_img[u] = in;
        unsigned int pix = kt_mad24(coo_y, (unsigned int)a.cols, coo_x);   // exact inside the image; selected away outside
        pix = in ? pix : 0u;
        rec[u] = a.rec[pix];
        r2[u] = __builtin_fmaf(e.vgz, e.vgz, v_g_part_norm);
        toff[u] = col_base2 + e.zoff2;
        v_x += dvx;  // the walk advances on every step, also on skipped ones
        v_y += dvy;
    }
    // ---- phase B: the update predicate (dp != 0 and sdf >= -trunc), the new tsdf sample, the volume words of the voxels that pass
    bool upd[KT_TSDF_UNROLL];
    float tv[KT_TSDF_UNROLL];
    unsigned int raw[KT_TSDF_UNROLL];   // the packed tsdf, zero-extended (one register each: packing two per register would wait for the loads here)
    unsigned int col[KT_TSDF_UNROLL];
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u) {
        const float Dp_scaled = fabsf(rec[u].dp);   // a negative scaled depth flags "no colour" (tsdf_volume.cu:520-527, :590-594)
        // v_sqrt_f32 (<= 1 ulp) gives t = sdf / trunc to within 2e-5.  t > 1.001: free space, min(1, t) is exactly 1 whatever the last bit
        // of the root.  t < -1.001: sdf < -trunc, no update.  In between (the truncation band and a hair around its edges) the correctly
        // rounded root runs, behind a wave-uniform branch.
        float sdf = Dp_scaled - __builtin_amdgcn_sqrtf(r2[u]);
        const float t = sdf * tranc_dist_inv;
        const bool live = in_img[u] & (rec[u].dp != 0);
        if constexpr (!TOL) {
            const bool band = live & (t <= 1.001f) & (t >= -1.001f);   // NaN (never: r2 >= 0) falls out
            if (__builtin_amdgcn_ballot_w64(band) != 0) {
                asm volatile("; exact sqrt" ::: "memory");  // a real branch: if-converted, the 16-instruction correctly rounded sqrt runs for every voxel
                if (band) sdf = Dp_scaled - __builtin_sqrtf(r2[u]);
            }
        }
        upd[u] = live & (sdf >= -a.tranc_dist);   // free space: sdf > 1.001 trunc; behind the band: sdf < -1.001 trunc -- the approximate root decides both
        tv[u] = kt_min1(sdf * tranc_dist_inv);
        if (COUNT && in_img[u]) ++n_img;
        if (upd[u]) {
            raw[u] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(m.vol, toff[u], 0, NT ? 2 : KT_TSDF_LD_AUX);
            col[u] = __builtin_amdgcn_raw_buffer_load_b32(m.col, kt_twice(toff[u]), 0, NT ? 2 : KT_TSDF_LD_AUX);
        }
    }
    // ---- phase C: running averages, stores (only of words that changed)
#pragma unroll
    for (int u = 0; u < KT_TSDF_UNROLL; ++u) {
        if (!upd[u]) continue;
        if (COUNT) ++n_upd;
        const unsigned int c = col[u];
        const float weight_prev = kt_ubyte_f32<3>(c);
        // a voxel that already holds F = 1 (raw 32767) and sees tsdf = 1: (1 * W + 1) / (W + 1) == 1 exactly, the stored value stays
        const bool touch = !((tv[u] == 1.0f) & (raw[u] == (unsigned int)KT_DIVISOR));
        if (touch) {
            const float tsdf_prev = SOL ? (float)(short)raw[u] * (1.0f / 32767.0f) : kt_unpack_tsdf((short)raw[u]);
            // (F W + tsdf) / (W + 1), correctly rounded: y = RN(1 / (W + 1)) from the LDS table, q = RN(n y), one exact residual, one
            // correction (Markstein); kt_debug_div_check compares it with the division for every finite numerator and divisor 1..256
            const float num = __builtin_fmaf(tsdf_prev, weight_prev, tv[u]), den = weight_prev + 1.0f;
            const float y = *(const float*)((const char*)s_rcp + ((c >> 22) & 0x3fcu));   // s_rcp[c >> 24]
            const float q0 = num * y;
            const short packed = kt_pack_tsdf(TOL ? q0 : __builtin_fmaf(__builtin_fmaf(-den, q0, num), y, q0));
            (void)den;
            if ((unsigned int)(unsigned short)packed != raw[u]) {   // an unchanged word is not written back
                __builtin_amdgcn_raw_buffer_store_b16(packed, m.vol, toff[u], 0, NT ? 2 : KT_TSDF_ST_AUX);
                if (a.bricks && packed < 0) a.bricks[s_tab[zb + u].bz + brick_xy] = 1;  // idempotent byte store (rare: the table is re-read here)
            }
        }
        // weight: min(W + 1, 128) in the top byte, on the whole word (saturating add: W = 255 must not wrap)
        unsigned int o = min(__builtin_elementwise_add_sat(c, 0x01000000u), (c & 0x00ffffffu) | 0x80000000u);
        const unsigned int rgbf = rec[u].rgbf;
        // colour update iff (normal valid and not flagged "no colour") or the stored colour is (0, 0, 0)  (tsdf_volume.cu:623); a voxel
        // whose stored colour already equals the pixel's keeps it (see kt_tsdf_consume)
        const bool blend = ((((rgbf & KT_REC_NORMAL_NAN) == 0) & (rec[u].dp > 0.0f)) | ((c & 0xffffffu) == 0)) & (((c ^ rgbf) & 0xffffffu) != 0);
        if (blend) {
            const float Wrkc = rec[u].wrkc;
            const float den = weight_prev + Wrkc;
            // numerators c_prev * W + Wrkc * c_new: the SECOND product is the fused one (oracle/_ref pins it at a .5 tie of the quotient)
            // (the products c_prev * W are exact in float: two integers below 256)
            const float nx = __builtin_fmaf(Wrkc, kt_ubyte_f32<0>(rgbf), kt_ubyte_f32<0>(c) * weight_prev);
            const float ny = __builtin_fmaf(Wrkc, kt_ubyte_f32<1>(rgbf), kt_ubyte_f32<1>(c) * weight_prev);
            const float nz = __builtin_fmaf(Wrkc, kt_ubyte_f32<2>(rgbf), kt_ubyte_f32<2>(c) * weight_prev);
            const float rden = __builtin_amdgcn_rcpf(den);
            // candidates k = the integer nearest n * rden, formed as MAGIC + k in one FMA (any integer next to the quotient will do: the
            // residual test below is what licenses it)
            float yx = __builtin_fmaf(nx, rden, KT_RNE_MAGIC), yy = __builtin_fmaf(ny, rden, KT_RNE_MAGIC), yz = __builtin_fmaf(nz, rden, KT_RNE_MAGIC);
            if constexpr (!TOL) {
                const float kx = yx - KT_RNE_MAGIC, ky = yy - KT_RNE_MAGIC, kz = yz - KT_RNE_MAGIC;
                const float lim = 0.4999f * den;
                // candidate accepted when the FMA residual |n - k den| < 0.4999 den (then the exact quotient rounds to k); false for NaN
                const bool safe = (fabsf(__builtin_fmaf(-kx, den, nx)) < lim) & (fabsf(__builtin_fmaf(-ky, den, ny)) < lim) &
                                  (fabsf(__builtin_fmaf(-kz, den, nz)) < lim);
                if (__builtin_amdgcn_ballot_w64(!safe) != 0) {
                    asm volatile("; exact blend" ::: "memory");
                    if (!safe) {
                        yx = (float)min(255, max(0, kt_f2i_rn(nx / den))) + KT_RNE_MAGIC;
                        yy = (float)min(255, max(0, kt_f2i_rn(ny / den))) + KT_RNE_MAGIC;
                        yz = (float)min(255, max(0, kt_f2i_rn(nz / den))) + KT_RNE_MAGIC;
                    }
                }
            }
            // (TOL: the candidates are taken as they are.  Quotients lie in [0, 255], so a candidate is 0..255 or, next to 255.5, never 256:
            // n <= 255 den.  den = 0 -- weight 0 and Wrkc 0 -- makes n * rden = 0 * inf the default NaN, whose low byte is 0: what the exact
            // kernel's rn(NaN) stores too.)
            // k in 0..255 sits in the low byte of MAGIC + k
            o = (o & 0xff000000u) | (__float_as_uint(yx) & 0xffu) | ((__float_as_uint(yy) & 0xffu) << 8) | ((__float_as_uint(yz) & 0xffu) << 16);
        }
        if (o != c) __builtin_amdgcn_raw_buffer_store_b32(o, m.col, kt_twice(toff[u]), 0, NT ? 2 : KT_TSDF_ST_AUX);   // a saturated free-space voxel in front of an unchanged pixel costs reads only
    }
}

// NT: the volume words are loaded and stored with the non-temporal policy (buffer aux bit 1).  A launch property: on a dense view the
// updated voxels (816 MB at 1280x960 into 768^3) stream through the caches once per frame, and marking them non-temporal keeps the pixel
// records -- which ARE re-read, by every z-step of every column -- in the L2s: 21 % fewer bytes fetched at the same launch time
// (profiles/r04_experiments.md).  On a sparse view the words a frame updates (20 MB) are still cached from the frame before, and
// non-temporal accesses give those hits up (orbit: 47 us against 32).
#ifdef KT_TSDF_TIMELINE   // analysis builds (scripts/tsdf_timeline.py): per wave {hw id, entry, tables ready, first batch, ..., exit} in 10 ns ticks
#define KT_TL_WORDS 16
__device__ unsigned long long kt_tsdf_tl[KT_TSDF_WAVES * KT_TL_WORDS];
#define KT_TL(i) do { if ((threadIdx.x & 63) == 0 && tl_n < KT_TL_WORDS) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KT_TL(i) do {} while (0)
#endif
// FP: the pose and the parked flag come from the device (kt_frame_params, the tracker) -- a compile-time property so that the start-up
// has no pointer test in front of its loads.
template <bool COUNT, bool NT, bool FP, int CT>
__device__ __forceinline__ void kt_tsdf23_lean_body(const kt_tsdf_lean_args& a_in)
{
#ifdef KT_TSDF_TIMELINE
    unsigned long long* tl = &kt_tsdf_tl[(size_t)(blockIdx.x * KT_TSDF_WPB + (threadIdx.x >> 6)) * KT_TL_WORDS];
    int tl_n = 0;
    if ((threadIdx.x & 63) == 0) { unsigned int hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); tl[tl_n++] = hw; for (int q = 1; q < KT_TL_WORDS; ++q) tl[q] = 0; }
    KT_TL(1);
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char kt_tsdf_lds[];
    float* s_rcp = (float*)kt_tsdf_lds;                                   // [256] RN(1 / (W + 1)) for every weight byte W
    kt_tsdf_ztab* s_tab = (kt_tsdf_ztab*)(kt_tsdf_lds + 1024);            // [N + KT_TSDF_UNROLL]
    const kt_tsdf_lean_args& a = a_in;
    // Start-up, in as few dependent memory round trips as the data allow (the in-kernel timeline of round 4, profiles/r04_experiments.md:
    // a wave spent 2.2 us before its tables were ready -- the parked-frame flag was tested BEFORE anything else was requested -- and 3.5 us
    // more on task_count -> task word -> {wave-column range, walk checkpoint}; every wave of a one-task-per-wave launch is in that phase
    // at the same time, so nothing hides it).  Now: trip 1 = {parked flag, pose, this XCD's part of the list, this thread's table entries},
    // trip 2 = the wave's first task word, under the table's LDS writes and the barrier; trip 3 = range + checkpoint.
    const int N = a.N;
    const unsigned int xcd = blockIdx.x & 7u;
    int parked = 0;
    float Ri[9], tx, ty, tz;   // the pose: from the device (kt_frame_params, written by the frame's set-up kernel) or from the arguments
    if constexpr (FP) {
        // t[3], Rinv[9] and skip are 13 consecutive words of kt_frame_params: requested together, one wait
        struct fp_tail { float t[3], Rinv[9]; int skip; };
        static_assert(offsetof(kt_frame_params, skip) == offsetof(kt_frame_params, t) + 12 * sizeof(float), "kt_frame_params tail");
        const fp_tail ft = *(const fp_tail*)&a.fp->t[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) Ri[k] = ft.Rinv[k];
        tx = ft.t[0]; ty = ft.t[1]; tz = ft.t[2];
        parked = ft.skip;
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) Ri[k] = a.Ri.m[k];
        tx = a.tx; ty = a.ty; tz = a.tz;
    }
    // XCD-aware task order (see kt_tsdf23_kernel): XCD k takes the k-th contiguous part of the list.
    // A parked frame does nothing here (its in-stream pre-pass left an empty task list, but a plan made ahead of the frame did not): its
    // part of the list is taken as empty.  Deliberately NOT an early return: a branch on the flag in front of everything else is what
    // made the flag's round trip the first of five.
    uint2 t_part;   // two adjacent words, only 4-byte aligned (odd word index for even xcd): copied, not read through a uint2 lvalue
    __builtin_memcpy(&t_part, &a.task_count[1 + xcd], sizeof(t_part));
    const unsigned int t_begin = t_part.x, t_end = parked != 0 ? t_part.x : t_part.y;
    constexpr int KT_TB = 64 * KT_TSDF_WPB;
    constexpr int KT_TAB_PASSES = (1023 + KT_TSDF_UNROLL + KT_TB - 1) / KT_TB;   // N < 1024 on this path
    float tab_vgz[KT_TAB_PASSES], tab_zs[KT_TAB_PASSES];
#pragma unroll
    for (int i = 0; i < KT_TAB_PASSES; ++i) {
        const int z = (int)threadIdx.x + KT_TB * i;
        const int zz = min(z, N - 1);   // entries past the volume are only ever read by masked-off steps; they repeat the last one
        tab_vgz[i] = a.vgz[zz]; tab_zs[i] = a.zs[zz];
    }
    const unsigned int t_stride = (gridDim.x >> 3) * (unsigned int)KT_TSDF_WPB;
    unsigned int t = t_begin + (blockIdx.x >> 3) * (unsigned int)KT_TSDF_WPB + (threadIdx.x >> 6);
    unsigned int task_word = t < t_end ? a.tasks[t] : 0u;
    const unsigned int plane = (unsigned int)N * (unsigned int)N;
    for (int w = (int)threadIdx.x; w < 256; w += KT_TB) s_rcp[w] = 1.0f / (float)(w + 1);
#pragma unroll
    for (int i = 0; i < KT_TAB_PASSES; ++i) {
        const int z = (int)threadIdx.x + KT_TB * i;
        if (z < N + KT_TSDF_UNROLL) {
            const int zz = min(z, N - 1);
            int sz = zz + a.wz; if (sz >= N) sz -= N;
            kt_tsdf_ztab e;
            e.vgz = tab_vgz[i]; e.zs = tab_zs[i];
            e.zoff2 = (unsigned int)sz * plane * 2u;
            e.bz = (sz >> KT_BRICK_LOG2) * a.nb * a.nb;
            s_tab[z] = e;
        }
    }
    __syncthreads();
    KT_TL(2);
    kt_tsdf_bufs m;
    {
        const unsigned int nvox = (unsigned int)a_in.N * (unsigned int)a_in.N * (unsigned int)a_in.N;
        m.vol = __builtin_amdgcn_make_buffer_rsrc((void*)a_in.volume, 0, nvox * 2u, 0x00020000);
        m.col = __builtin_amdgcn_make_buffer_rsrc((void*)a_in.color, 0, nvox * 4u, 0x00020000);
    }
    const int WX = 1 << a.wcl, WY = 64 >> a.wcl;
    const int lane = threadIdx.x & 63;
    const float v_g_z0 = __builtin_fmaf(0 + 0.5f, a.cell_z, -tz);
    const float dvx = Ri[2] * a.cell_z * a.intr.fx;   // Rcurr_inv_0_z_scaled
    const float dvy = Ri[5] * a.cell_z * a.intr.fy;   // Rcurr_inv_1_z_scaled
    const float tranc_dist_inv = 1.0f / a.tranc_dist;
    const float r8 = Ri[8];
    unsigned int n_upd = 0, n_batches = 0, n_tasks_done = 0, n_img = 0;
    for (; t < t_end; t += t_stride, task_word = t < t_end ? a.tasks[t] : 0u) {
        const unsigned int task = __builtin_amdgcn_readfirstlane(task_word);
        const int yg = (int)(task & 0xffffu), xg = (int)((task >> 16) & 0xffu), chunk = (int)(task >> 24);
        const int sx = xg * WX + (lane & (WX - 1));
        const int sy = min(yg * WY + (lane >> a.wcl), N - 1);   // a row past the volume (odd N) only repeats the last one, never live
        const bool lane_ok = (sx < N) & (yg * WY + (lane >> a.wcl) < N);
        const unsigned int wr = __builtin_amdgcn_readfirstlane(a.wrange[(size_t)yg * ((N + WX - 1) / WX) + xg]);
        const int zc = (int)(wr & 0xffffu);
        const int wz0 = max(zc, chunk * KT_TSDF_ZCHUNK), wz1 = min((int)(wr >> 16), (chunk + 1) * KT_TSDF_ZCHUNK);
        if (wz0 >= wz1) continue;
        int x = sx - a.wx; if (x < 0) x += N;
        int y = sy - a.wy; if (y < 0) y += N;
        const float v_g_x = __builtin_fmaf((float)x + 0.5f, a.cell_x, -tx);
        const float v_g_y = __builtin_fmaf((float)y + 0.5f, a.cell_y, -ty);
        const float v_g_part_norm = __builtin_fmaf(v_g_x, v_g_x, v_g_y * v_g_y);
        const float v_z = __builtin_fmaf(Ri[8], v_g_z0, __builtin_fmaf(Ri[6], v_g_x, Ri[7] * v_g_y));
        // the reference's walk (tsdf_volume.cu:566-571, 634-640): resume from the wave-column's checkpoint and advance to this task's first z
        const float2 cp = a.walk0[(size_t)sy * N + min(sx, N - 1)];
        float v_x = cp.x, v_y = cp.y;
        {
            int z = zc;
            for (; z + 16 <= wz0; z += 16) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { v_x += dvx; v_y += dvy; }
            }
            for (; z < wz0; ++z) { v_x += dvx; v_y += dvy; }
        }
        const unsigned int col_base = (unsigned int)min(sx, N - 1) + (unsigned int)sy * (unsigned int)N;
        const int brick_xy = (sy >> KT_BRICK_LOG2) * a.nb + (min(sx, N - 1) >> KT_BRICK_LOG2);
        // d(z) = fma(R8, z_scaled(z), v_z) is monotone in z: when its values at the two ends of the task are of one sign and in
        // [2^-20, 2^20] in magnitude, every reciprocal of the task may use the unwrapped refinement chain (kt_rcp_exact), and
        // "inv_z < 0" is the sign of d at either end: it joins the task's lane mask.  Otherwise the task takes the division.
        const int z_last = min(wz0 + ((wz1 - wz0 + KT_TSDF_UNROLL - 1) & ~(KT_TSDF_UNROLL - 1)) - 1, N - 1);
        const float d_a = __builtin_fmaf(r8, s_tab[wz0].zs, v_z), d_b = __builtin_fmaf(r8, s_tab[z_last].zs, v_z);
        const bool d_ok = fminf(fabsf(d_a), fabsf(d_b)) >= 0x1p-20f && fmaxf(fabsf(d_a), fabsf(d_b)) <= 0x1p20f && (d_a < 0) == (d_b < 0);
        const bool fast = CT == 2 ? __builtin_amdgcn_ballot_w64((d_a < 0) != (d_b < 0)) == 0 : __builtin_amdgcn_ballot_w64(!d_ok) == 0;   // (speed of light: no range test)
        KT_TL(3);
        if (fast) {
            if (lane_ok & (d_a > 0)) {   // lanes behind the camera plane (1 / d < 0) never pass the in-image test
                for (int zb = wz0; zb < wz1; zb += KT_TSDF_UNROLL) {
                    kt_tsdf_batch_lean<COUNT, true, NT, CT>(a, m, s_tab, s_rcp, zb, wz1 - zb, col_base * 2u, brick_xy, v_z, v_x, v_y, dvx, dvy, r8, v_g_part_norm, tranc_dist_inv, n_upd, n_img);
                    KT_TL(4);
                }
            }
        } else if (lane_ok) {
            for (int zb = wz0; zb < wz1; zb += KT_TSDF_UNROLL) {
                kt_tsdf_batch_lean<COUNT, false, NT, CT>(a, m, s_tab, s_rcp, zb, wz1 - zb, col_base * 2u, brick_xy, v_z, v_x, v_y, dvx, dvy, r8, v_g_part_norm, tranc_dist_inv, n_upd, n_img);
            }
        }
        if (COUNT) { n_batches += (unsigned int)((wz1 - wz0 + KT_TSDF_UNROLL - 1) / KT_TSDF_UNROLL); ++n_tasks_done; }
    }
    KT_TL(9);
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { n_upd += __shfl_down(n_upd, off, 64); n_img += __shfl_down(n_img, off, 64); }
        if (lane == 0) {
            if (n_upd) atomicAdd(a.updated, n_upd);
            if (n_batches) atomicAdd(a.updated + 1, n_batches);
            if (n_tasks_done) atomicAdd(a.updated + 2, n_tasks_done);
            if (n_img) atomicAdd(a.updated + 3, n_img);
        }
    }
}
// the two contracts as two kernels, so that a profile names the one that ran
template <bool COUNT, bool NT, bool FP>
__global__ __launch_bounds__(64 * KT_TSDF_WPB, COUNT ? 6 : KT_TSDF_OCC) void kt_tsdf23_lean_kernel(const kt_tsdf_lean_args a_in)
{
    kt_tsdf23_lean_body<COUNT, NT, FP, 0>(a_in);
}
template <bool COUNT, bool NT, bool FP>
__global__ __launch_bounds__(64 * KT_TSDF_WPB, COUNT ? 6 : KT_TSDF_OCC) void kt_tsdf23_tol_kernel(const kt_tsdf_lean_args a_in)
{
    kt_tsdf23_lean_body<COUNT, NT, FP, 1>(a_in);
}
template <bool COUNT, bool NT, bool FP>
__global__ __launch_bounds__(64 * KT_TSDF_WPB, COUNT ? 6 : KT_TSDF_OCC) void kt_tsdf23_sol_kernel(const kt_tsdf_lean_args a_in)   // measurement only (see kt_tsdf_batch_lean)
{
    kt_tsdf23_lean_body<COUNT, NT, FP, 2>(a_in);
}

// scratch owned by the context for integrate (pixel records, z tables, intervals, task list), grown on demand
struct kt_integrate_scratch {
    kt_pixrec* rec = nullptr; size_t rec_px = 0;
    float* vgz = nullptr; float* zs = nullptr; int tabN = 0;
    float* tab_host[2] = {nullptr, nullptr};  // pinned staging of {vgz[N], zs[N]}, double-buffered
    unsigned int* wrange = nullptr;            // N * ceil(N / 64) wave-column unions
    float2* walk0 = nullptr;                   // N * N walk checkpoints at the wave-column's first z
    float* dpmax = nullptr;                    // KT_DPT_FLOATS: tile maxima of |scaled depth|, two levels (non-prepared path)
    unsigned int* tasks = nullptr;             // up to N * ceil(N / 64) * ceil(N / ZCHUNK) tasks
    unsigned int* task_count = nullptr;
    int flip = 0;
};

void kt_integrate_scratch_free(kt_ctx* c)
{
    kt_integrate_scratch* s = c->integ;
    if (!s) return;
    (void)hipFree(s->rec); (void)hipFree(s->vgz); (void)hipFree(s->wrange); (void)hipFree(s->tasks); (void)hipFree(s->walk0); (void)hipFree(s->dpmax);
    (void)hipFree(s->task_count);
    for (int k = 0; k < 2; ++k) (void)hipHostFree(s->tab_host[k]);
    delete s;
    c->integ = nullptr;
}

static int kt_integrate_scratch_reserve(kt_ctx* c, size_t px, int N)
{
    if (!c->integ) c->integ = new kt_integrate_scratch();
    kt_integrate_scratch& s = *c->integ;
    if (s.rec_px < px) {
        KT_HIP(hipStreamSynchronize(c->stream));
        if (s.rec) KT_HIP(hipFree(s.rec));
        (void)hipFree(s.dpmax);
        s.dpmax = nullptr;
        KT_HIP(hipMalloc((void**)&s.dpmax, sizeof(float) * (size_t)KT_DPT_FLOATS));
        s.rec = nullptr; s.rec_px = 0;
        KT_HIP(hipMalloc((void**)&s.rec, px * sizeof(kt_pixrec)));
        s.rec_px = px;
    }
    if (s.tabN < N) {
        KT_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(s.vgz); (void)hipFree(s.wrange); (void)hipFree(s.tasks); (void)hipFree(s.task_count); (void)hipFree(s.walk0);
        s.vgz = s.zs = nullptr; s.wrange = s.tasks = s.task_count = nullptr; s.walk0 = nullptr; s.tabN = 0;
        const size_t wave_cols = kt_tsdf_max_wave_cols(N);
        KT_HIP(hipMalloc((void**)&s.wrange, sizeof(unsigned int) * wave_cols));
        KT_HIP(hipMalloc((void**)&s.walk0, sizeof(float2) * (size_t)N * N));
        KT_HIP(hipMalloc((void**)&s.tasks, sizeof(unsigned int) * wave_cols * kt_div_up(N, KT_TSDF_ZCHUNK)));
        KT_HIP(hipMalloc((void**)&s.task_count, sizeof(unsigned int) * kt_task_head_words(N)));
        KT_HIP(hipMalloc((void**)&s.vgz, sizeof(float) * 2 * N));
        s.zs = s.vgz + N;
        for (int k = 0; k < 2; ++k) {
            if (s.tab_host[k]) KT_HIP(hipHostFree(s.tab_host[k]));
            KT_HIP(hipHostMalloc((void**)&s.tab_host[k], sizeof(float) * 2 * N, hipHostMallocDefault));
        }
        s.tabN = N;
    }
    return KT_OK;
}

// the two pre-pass launches: column intervals (+ wave-column unions, walk checkpoints unless walk0 is null), then the task list
static int kt_tsdf_prepass(hipStream_t stream, const kt_tsdf23_args& a, unsigned int* wrange, float2* walk0, unsigned int* tasks, unsigned int* task_count)
{
    const int N = a.N;
    const int WX = 1 << a.wcl, WY = 64 >> a.wcl;
    const int XG = kt_div_up(N, WX), YG = kt_div_up(N, WY);
    const size_t maps_lds = a.dpmax && a.dpt_log2 ? sizeof(float) * (size_t)(((kt_div_up(a.cols, 1 << a.dpt_log2) * kt_div_up(a.rows, 1 << a.dpt_log2) + 3) & ~3) + kt_div_up(a.cols, 32) * kt_div_up(a.rows, 32)) : 0;
    hipLaunchKernelGGL(kt_tsdf_interval_kernel, dim3(kt_div_up(N, 2 * WX), kt_div_up(N, 2 * WY)), dim3(256), maps_lds, stream, a, wrange, walk0);
    KT_LAUNCH_CHECK();
    static const bool serial = getenv("KT_TSDF_TASKS_SERIAL") != nullptr;   // (A/B switch: round 3's one-workgroup list kernel)
    if (!serial) {
        hipLaunchKernelGGL(kt_tsdf_tasks_scan_kernel, dim3(1), dim3(1024), 0, stream, wrange, XG * YG, task_count);
        hipLaunchKernelGGL(kt_tsdf_tasks_place_kernel, dim3(kt_div_up((XG * YG + 1) / 2, 4)), dim3(256), 0, stream, wrange, XG * YG, XG, task_count, tasks);
    } else
        hipLaunchKernelGGL(kt_tsdf_tasks_kernel, dim3(1), dim3(1024), 0, stream, wrange, XG * YG, XG, tasks, task_count);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// ---- planning ahead -------------------------------------------------------------------------------------------------------------
// The pre-pass above costs 24 us of a 370 us frame and sits between the odometry and the voxel kernel.  Its output only SELECTS work
// (every kept voxel still runs the reference's exact in-image test), so it may be computed for any pose that is known to be close
// to the frame's: the tracker predicts the pose of frame k + 1 from those of frames k and k - 1, runs the pre-pass for that
// prediction on a side stream while the odometry of frame k + 1 iterates, widened by margins (theta, tau) on the rotation and
// translation error, and the frame's set-up kernel checks the pose the odometry arrived at against those margins: inside them the
// voxel kernel runs from the plan, outside them the frame is parked and fused through the in-stream pre-pass instead.
int kt_tsdf_plan_alloc(kt_tsdf_plan* p, int N)
{
    const size_t wave_cols = kt_tsdf_max_wave_cols(N);
    memset(p, 0, sizeof(*p));
    KT_HIP(hipMalloc((void**)&p->wrange, sizeof(unsigned int) * wave_cols));
    KT_HIP(hipMalloc((void**)&p->walk0, sizeof(float2) * (size_t)N * N));
    KT_HIP(hipMalloc((void**)&p->tasks, sizeof(unsigned int) * wave_cols * kt_div_up(N, KT_TSDF_ZCHUNK)));
    KT_HIP(hipMalloc((void**)&p->task_count, sizeof(unsigned int) * kt_task_head_words(N)));
    return KT_OK;
}
void kt_tsdf_plan_free(kt_tsdf_plan* p)
{
    (void)hipFree(p->wrange); (void)hipFree(p->walk0); (void)hipFree(p->tasks); (void)hipFree(p->task_count);
    memset(p, 0, sizeof(*p));
}
void kt_tsdf_plan_shape(int cols, int rows, int N, int* wx, int* wy, int* xg, int* yg)
{
    const int wcl = kt_tsdf_wcl(cols, rows, N);
    *wx = 1 << wcl; *wy = 64 >> wcl; *xg = kt_div_up(N, *wx); *yg = kt_div_up(N, *wy);
}

int kt_integrate_plan(hipStream_t stream, const kt_tsdf_plan* plan, const void* rec, const float* dpmax, int cols, int rows, const kt_intr* intr,
                      const float volume_size[3], const kt_mat33* Rinv_pred, const float t_pred[3], float tranc_dist, const int voxel_wrap[3], int N,
                      float theta, float tau)
{
    KT_ARG(plan && rec && intr && volume_size && Rinv_pred && t_pred && voxel_wrap && N > 0 && N <= 1536 && theta >= 0 && tau >= 0);
    kt_tsdf23_args a;
    memset(&a, 0, sizeof(a));
    a.rec = (const kt_pixrec*)rec;
    a.Ri = *Rinv_pred;
    a.tx = t_pred[0]; a.ty = t_pred[1]; a.tz = t_pred[2];
    a.intr = *intr;
    a.cell_x = volume_size[0] / N; a.cell_y = volume_size[1] / N; a.cell_z = volume_size[2] / N;
    a.tranc_dist = tranc_dist;
    a.wx = voxel_wrap[0] % N; a.wy = voxel_wrap[1] % N; a.wz = voxel_wrap[2] % N;
    a.cols = cols; a.rows = rows; a.N = N;
    a.dpmax = dpmax;
    a.dpt_log2 = kt_dpt_log2(cols, rows);
    a.wcl = kt_tsdf_wcl(cols, rows, N);
    // |p| / p_z of a point that projects into the (padded) image is at most kappa; eps <= (theta kappa p_z + tau (1 + theta)) / (1 - theta kappa)
    const float kx = (fmaxf(intr->cx, (float)cols - intr->cx) + 3.0f) / intr->fx, ky = (fmaxf(intr->cy, (float)rows - intr->cy) + 3.0f) / intr->fy;
    const float kappa = sqrtf(1.0f + kx * kx + ky * ky);
    if (!(theta * kappa < 0.25f)) return KT_NO_PLAN;   // the margin algebra needs theta kappa << 1: no plan, not an error (the caller falls back to the in-stream pre-pass)
    a.pm_A = 1.01f * theta * kappa / (1.0f - theta * kappa);
    a.pm_B = 1.01f * tau * (1.0f + theta) / (1.0f - theta * kappa);
    return kt_tsdf_prepass(stream, a, plan->wrange, nullptr, plan->tasks, plan->task_count);
}

// test / A-B hook: which voxel kernel the N < 1024 launches use (-1: KT_TSDF_LEAN in the environment, else the build's default)
static int kt_tsdf_lean_override = -1;
extern "C" int kt_debug_tsdf_lean(int on) { kt_tsdf_lean_override = on < 0 ? -1 : (on != 0); return KT_OK; }
static bool kt_tsdf_lean_selected()
{
    static const bool lean_env = []() { const char* e = getenv("KT_TSDF_LEAN"); return e ? atoi(e) != 0 : KT_TSDF_LEAN_DEFAULT != 0; }();
    return kt_tsdf_lean_override < 0 ? lean_env : kt_tsdf_lean_override != 0;
}
extern "C" int kt_debug_tsdf_timeline(kt_ctx* c, unsigned long long* out_host, int max_words)
{
    KT_ARG(c && out_host && max_words > 0);
#ifdef KT_TSDF_TIMELINE
    KT_HIP(hipStreamSynchronize(c->stream));
    const size_t n = (size_t)KT_TSDF_WAVES * KT_TL_WORDS;
    KT_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(kt_tsdf_tl), sizeof(unsigned long long) * (n < (size_t)max_words ? n : (size_t)max_words)));
    return (int)KT_TL_WORDS;   // (> 0: words per wave)
#else
    kt_set_error("kt_debug_tsdf_timeline: the library was not built with -DKT_TSDF_TIMELINE");
    return -KT_ERR_STATE;
#endif
}
// which arithmetic contract the lean voxel kernel runs under: 0 = bit-exact (the default), 1 = survey-8c (kt_tsdf23_tol_kernel, above);
// -1 = back to the environment's (KT_TSDF_CONTRACT=survey8c) or the default.  Only the lean kernel has the second contract.
// 2 = the speed-of-light measurement variant (kt_tsdf23_sol_kernel; KT_TSDF_CONTRACT=sol): results are NOT the reference's, bench.py only times it
static int kt_tsdf_contract_override = -1;
extern "C" int kt_debug_tsdf_contract(int tol) { kt_tsdf_contract_override = tol < 0 ? -1 : (tol > 2 ? 1 : tol); return KT_OK; }
static int kt_tsdf_contract_selected()
{
    static const int env = []() {
        const char* e = getenv("KT_TSDF_CONTRACT");
        if (!e) return 0;
        if (!strcmp(e, "survey8c") || !strcmp(e, "survey-8c") || !strcmp(e, "tol")) return 1;
        return (!strcmp(e, "sol") || !strcmp(e, "speed-of-light")) ? 2 : 0;
    }();
    return kt_tsdf_lean_selected() ? (kt_tsdf_contract_override < 0 ? env : kt_tsdf_contract_override) : 0;
}
extern "C" const char* kt_debug_tsdf_kernel(void)
{
    static const char* const names[3] = {"kt_tsdf23_lean_kernel", "kt_tsdf23_tol_kernel", "kt_tsdf23_sol_kernel"};
    return kt_tsdf_lean_selected() ? names[kt_tsdf_contract_selected()] : "kt_tsdf23_kernel";
}

// shared by the C entry point and the tracker (which wants the update count for the roofline report)
int kt_integrate_tsdf_impl(kt_ctx* c, const uint16_t* depth_raw, int cols, int rows, const kt_intr* intr,
                           const float volume_size[3], const kt_mat33* Rcurr_inv, const float tcurr[3], float tranc_dist,
                           int16_t* volume, float* depth_raw_scaled, const int voxel_wrap[3], uint8_t* color_volume,
                           const uint8_t* colors, const float* nmap_curr, int angle_color, int N, unsigned int* updated_dev,
                           const void* prepared_rec, const kt_frame_params* fp, unsigned char* bricks, const float* prepared_dpmax,
                           const kt_tsdf_plan* plan)
{
    KT_ARG(c && depth_raw && intr && volume_size && Rcurr_inv && tcurr && volume && depth_raw_scaled && voxel_wrap &&
           color_volume && colors && nmap_curr && N > 0 && cols > 0 && rows > 0);
    int s = kt_integrate_scratch_reserve(c, (size_t)cols * rows, N);
    if (s != KT_OK) return s;
    const float cell_x = volume_size[0] / N, cell_y = volume_size[1] / N, cell_z = volume_size[2] / N;
    // the incremental z walk of tsdf23 (quirk A.17) is the same float sequence for every column: build it once on the
    // host (plain IEEE float adds, this file is compiled with -ffp-contract=off) and ship 2 * N floats with the frame
    if (!fp) {
        kt_integrate_scratch& sc = *c->integ;
        float* th = sc.tab_host[sc.flip];
        sc.flip ^= 1;
        float v_g_z = fmaf(0 + 0.5f, cell_z, -tcurr[2]);
        float z_scaled = 0;
        for (int z = 0; z < N; ++z) {
            th[z] = v_g_z;
            th[sc.tabN + z] = z_scaled;
            v_g_z += cell_z;
            z_scaled += cell_z;
        }
        KT_HIP(hipMemcpyAsync(sc.vgz, th, sizeof(float) * N, hipMemcpyHostToDevice, c->stream));
        KT_HIP(hipMemcpyAsync(sc.zs, th + sc.tabN, sizeof(float) * N, hipMemcpyHostToDevice, c->stream));
    }
    if (!prepared_rec) {  // scaleDepth + per-pixel records (a caller that ran kt_integrate_prepare ahead of time passes them in)
        kt_launch_scale_depth(c, depth_raw, depth_raw_scaled, c->integ->rec, colors, nmap_curr, cols, rows, *intr, angle_color, c->integ->dpmax);
        KT_LAUNCH_CHECK();
        prepared_dpmax = kt_dpt_log2(cols, rows) ? c->integ->dpmax : nullptr;
    }
    kt_tsdf23_args a;
    a.rec = prepared_rec ? (const kt_pixrec*)prepared_rec : c->integ->rec;
    a.volume = volume;
    a.color = (uchar4*)color_volume;
    a.vgz = c->integ->vgz;
    a.zs = c->integ->zs;
    a.updated = updated_dev;
    a.fp = fp;
    a.nb = N / KT_BRICK;
    a.bricks = (bricks && (N % KT_BRICK) == 0) ? bricks : nullptr;
    a.Ri = *Rcurr_inv;
    a.tx = tcurr[0]; a.ty = tcurr[1]; a.tz = tcurr[2];
    a.intr = *intr;
    a.cell_x = cell_x; a.cell_y = cell_y; a.cell_z = cell_z;
    a.tranc_dist = tranc_dist;
    for (int k = 0; k < 3; ++k) KT_ARG(voxel_wrap[k] >= 0);  // vWrapCopy is always normalised (KintinuousTracker.cpp:1075-1085)
    a.wx = voxel_wrap[0] % N; a.wy = voxel_wrap[1] % N; a.wz = voxel_wrap[2] % N;
    a.cols = cols; a.rows = rows; a.N = N;
    KT_ARG(N <= 1536);  // 32-bit voxel offsets (N^3 < 2^32) and 16-bit z bounds
    a.pm_A = a.pm_B = 0.0f;
    a.dpmax = prepared_dpmax;
    a.dpt_log2 = kt_dpt_log2(cols, rows);
    a.wcl = kt_tsdf_wcl(cols, rows, N);
    if (plan) {
        // the task plan was made ahead of the frame (kt_integrate_plan, conservative for every pose within its margins -- the caller
        // has checked that this frame's pose is) and its walk checkpoints by the frame's set-up kernel: only the voxel kernel is left
        a.tasks = plan->tasks; a.task_count = plan->task_count; a.wrange = plan->wrange; a.walk0 = plan->walk0;
    } else {
        a.tasks = c->integ->tasks; a.task_count = c->integ->task_count; a.wrange = c->integ->wrange; a.walk0 = c->integ->walk0;
        KT_TRY(kt_tsdf_prepass(c->stream, a, c->integ->wrange, c->integ->walk0, c->integ->tasks, c->integ->task_count));
    }
    dim3 b(256), g(KT_TSDF_WAVES / 4);
    if (kt_tsdf23_hook.on) KT_HIP(hipEventRecord(kt_tsdf23_hook.ev[0], c->stream));
    const bool buf = N < 1024 && !getenv("KT_TSDF_POINTERS");   // 32-bit byte offsets into the colour volume (N^3 * 4 < 2^32)
    if (buf && kt_tsdf_lean_selected()) {   // round 4: the kernel with the scalar half cut (same stored bits; KT_TSDF_LEAN=0 selects the round-3 kernel)
        kt_tsdf_lean_args l;
        l.rec = a.rec; l.volume = a.volume; l.color = a.color; l.vgz = a.vgz; l.zs = a.zs; l.tasks = a.tasks; l.task_count = a.task_count;
        l.wrange = a.wrange; l.walk0 = a.walk0; l.updated = a.updated; l.fp = a.fp; l.bricks = a.bricks; l.Ri = a.Ri;
        l.tx = a.tx; l.ty = a.ty; l.tz = a.tz; l.intr = a.intr; l.cell_x = a.cell_x; l.cell_y = a.cell_y; l.cell_z = a.cell_z;
        l.tranc_dist = a.tranc_dist; l.wx = a.wx; l.wy = a.wy; l.wz = a.wz; l.cols = a.cols; l.rows = a.rows; l.N = a.N; l.nb = a.nb; l.wcl = a.wcl;
        const size_t lds = 1024 + sizeof(kt_tsdf_ztab) * (size_t)(N + KT_TSDF_UNROLL);
        // dense view (the rule that picks the 32 x 2 wave-column shape): the volume words stream, non-temporal; KT_TSDF_NT=0|1 overrides
        static const int nt_env = []() { const char* e = getenv("KT_TSDF_NT"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
        const bool nt = nt_env >= 0 ? nt_env != 0 : a.wcl == 5;
        const dim3 lb(64 * KT_TSDF_WPB), lg(KT_TSDF_WAVES / KT_TSDF_WPB);
        const int contract = kt_tsdf_contract_selected();
        const bool tol = contract == 1;
#define KT_LEAN_LAUNCH(C, T) do { \
            if (contract == 2) { if (l.fp) hipLaunchKernelGGL((kt_tsdf23_sol_kernel<C, T, true>), lg, lb, lds, c->stream, l); else hipLaunchKernelGGL((kt_tsdf23_sol_kernel<C, T, false>), lg, lb, lds, c->stream, l); } \
            else if (tol) { if (l.fp) hipLaunchKernelGGL((kt_tsdf23_tol_kernel<C, T, true>), lg, lb, lds, c->stream, l); else hipLaunchKernelGGL((kt_tsdf23_tol_kernel<C, T, false>), lg, lb, lds, c->stream, l); } \
            else { if (l.fp) hipLaunchKernelGGL((kt_tsdf23_lean_kernel<C, T, true>), lg, lb, lds, c->stream, l); else hipLaunchKernelGGL((kt_tsdf23_lean_kernel<C, T, false>), lg, lb, lds, c->stream, l); } } while (0)
        if (updated_dev) { if (nt) KT_LEAN_LAUNCH(true, true); else KT_LEAN_LAUNCH(true, false); }
        else { if (nt) KT_LEAN_LAUNCH(false, true); else KT_LEAN_LAUNCH(false, false); }
#undef KT_LEAN_LAUNCH
    } else if (updated_dev) {
        if (buf) hipLaunchKernelGGL((kt_tsdf23_kernel<true, true>), g, b, 0, c->stream, a);
        else hipLaunchKernelGGL((kt_tsdf23_kernel<true, false>), g, b, 0, c->stream, a);
    } else {
        if (buf) hipLaunchKernelGGL((kt_tsdf23_kernel<false, true>), g, b, 0, c->stream, a);
        else hipLaunchKernelGGL((kt_tsdf23_kernel<false, false>), g, b, 0, c->stream, a);
    }
    KT_LAUNCH_CHECK();
    if (kt_tsdf23_hook.on) {
        kt_tsdf23_hook.on = false;  // one-shot: armed by the tracker per call
        KT_HIP(hipEventRecord(kt_tsdf23_hook.ev[1], c->stream));
    }
    return KT_OK;
}

int kt_integrate_tables(kt_ctx* c, int cols, int rows, int N, float** vgz, float** zs)
{
    int s = kt_integrate_scratch_reserve(c, (size_t)cols * rows, N);
    if (s != KT_OK) return s;
    *vgz = c->integ->vgz;
    *zs = c->integ->zs;
    return KT_OK;
}

// The pose-independent half of integrateTsdfVolume (scaleDepth, tsdf_volume.cu:493-511, plus the per-pixel records): the tracker
// runs it for frame k + 1 on its prefetch stream while frame k is still being tracked.
size_t kt_integrate_rec_bytes(int cols, int rows) { return (size_t)cols * rows * sizeof(kt_pixrec); }
size_t kt_integrate_dpmax_bytes(int, int) { return sizeof(float) * (size_t)KT_DPT_FLOATS; }
int kt_integrate_prepare(kt_ctx* c, const uint16_t* depth_raw, const uint8_t* colors, const float* nmap_curr, int cols, int rows,
                         const kt_intr* intr, int angle_color, float* depth_raw_scaled, void* rec, float* dpmax)
{
    kt_launch_scale_depth(c, depth_raw, depth_raw_scaled, (kt_pixrec*)rec, colors, nmap_curr, cols, rows, *intr, angle_color, dpmax);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// [A] of processFrame behind the bilateral filter, as the tracker enqueues it: pyramid levels 0 / 1 (kt_pyramid01_kernel), then ONE launch for
// levels 2 / 3 and scaleDepth + pixel records (kt_prepare_fused_kernel), then the tile map's finishing workgroup.  Same outputs as
// kt_build_pyramid + kt_integrate_prepare, bit for bit (the same block functions).
int kt_frame_prepare(kt_ctx* c, const kt_intr* intr, const uint16_t* depth_filtered, uint16_t* const depths_out[3], float* const vmaps[4], float* const nmaps[4],
                     const uint16_t* depth_raw, const uint8_t* colors, int cols, int rows, int angle_color, float* depth_raw_scaled, void* rec, float* dpmax)
{
    KT_ARG(c && intr && depth_filtered && depths_out && vmaps && nmaps && depth_raw && colors && depth_raw_scaled && rec);
    kt_pyr_args pa;
    KT_TRY(kt_pyr_args_fill(&pa, intr, depth_filtered, cols, rows, depths_out, vmaps, nmaps));
    KT_TRY(kt_pyramid01_launch(c, &pa));
    const kt_sd_args sa = kt_sd_args_fill(depth_raw, depth_raw_scaled, (kt_pixrec*)rec, colors, nmaps[0], cols, rows, *intr, angle_color, dpmax);
    const int n23x = kt_div_up(cols / 8, 4), n23 = n23x * kt_div_up(rows / 8, 4);
    const int nsd = sa.strips_x * kt_div_up(rows, 8 * sa.passes);
    hipLaunchKernelGGL(kt_prepare_fused_kernel, dim3(n23 + nsd), dim3(256), 0, c->stream, pa, sa, n23x, n23);
    kt_launch_tile_finish(c, sa, dpmax);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

extern "C" int kt_integrate_tsdf(kt_ctx* c, const uint16_t* depth_raw, int cols, int rows, const kt_intr* intr,
                                 const float volume_size[3], const kt_mat33* Rcurr_inv, const float tcurr[3], float tranc_dist,
                                 int16_t* volume, float* depth_raw_scaled, const int voxel_wrap[3], uint8_t* color_volume,
                                 const uint8_t* colors, const float* nmap_curr, int angle_color, int N)
{
    return kt_integrate_tsdf_impl(c, depth_raw, cols, rows, intr, volume_size, Rcurr_inv, tcurr, tranc_dist, volume,
                                  depth_raw_scaled, voxel_wrap, color_volume, colors, nmap_curr, angle_color, N, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// exhaustive check hook for kt_rcp_exact: every float d with 2^-20 <= |d| <= 2^20 (both signs) against the IEEE division
__global__ __launch_bounds__(256) void kt_rcp_check_kernel(unsigned int* __restrict__ mismatches)
{
    const unsigned int lo = 0x35800000u, hi = 0x49800000u;   // 2^-20 .. 2^20
    unsigned int bad = 0;
    for (unsigned int bits = lo + blockIdx.x * 256u + threadIdx.x; bits <= hi; bits += gridDim.x * 256u) {
        const float d = __uint_as_float(bits);
        float q = 1.0f / d;
        asm volatile("" : "+v"(q));
        if (__float_as_uint(kt_rcp_exact(d)) != __float_as_uint(q)) ++bad;
        float qn = 1.0f / -d;
        asm volatile("" : "+v"(qn));
        if (__float_as_uint(kt_rcp_exact(-d)) != __float_as_uint(qn)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}
extern "C" int kt_debug_rcp_check(kt_ctx* c, unsigned int* mismatches_host)
{
    KT_ARG(c && mismatches_host);
    unsigned int* d = nullptr;
    KT_HIP(hipMalloc((void**)&d, sizeof(unsigned int)));
    KT_HIP(hipMemsetAsync(d, 0, sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL(kt_rcp_check_kernel, dim3(4096), dim3(256), 0, c->stream, d);
    KT_HIP(hipMemcpyAsync(mismatches_host, d, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    KT_HIP(hipFree(d));
    return KT_OK;
}

// exhaustive check hook for kt_unpack_tsdf (division by 32767 restated as a multiply and two FMAs): out[v + 32768] = unpack(v)
__global__ void kt_unpack_table_kernel(float* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 65536) out[i] = kt_unpack_tsdf((short)(i - 32768));
}
extern "C" int kt_debug_unpack_table(kt_ctx* c, float* out_host65536)
{
    KT_ARG(c && out_host65536);
    float* d = nullptr;
    KT_HIP(hipMalloc((void**)&d, 65536 * sizeof(float)));
    hipLaunchKernelGGL(kt_unpack_table_kernel, dim3(256), dim3(256), 0, c->stream, d);
    KT_HIP(hipMemcpyAsync(out_host65536, d, 65536 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    KT_HIP(hipFree(d));
    return KT_OK;
}

// ================================================================================================
// a12  raycast -> rayCastKernel                       ray_caster.cu:56-471
// ================================================================================================
struct kt_raycast_args {
    kt_mat33 R;
    float tx, ty, tz;
    float time_step;
    float vsx, vsy, vsz;        // volume size
    float cx_, cy_, cz_;        // cell size
    int cols, rows, N;
    const int16_t* volume;
    const uchar4* color;
    kt_intr intr;
    float* vmap; float* nmap;
    int wx, wy, wz;
    uchar4* vmap_color;
    unsigned long long* steps;  // optional march-step counter (S of SURVEY 8d)
    // optional fused resizeVMap / resizeNMap outputs for levels 1..3 (maps.cu:225-308), tracker path only
    float* vpyr[3]; float* npyr[3];
    const kt_frame_params* fp;  // when set: R / t come from the device
    const unsigned char* bricks; int nb;   // optional negative-brick flags maintained by tsdf23 (N % 32 == 0, nb^3 <= KT_RC_MAX_BRICKS)
};

struct kt_rc {
    const kt_raycast_args& a;
    float rcx, rcy, rcz;   // RN(1 / cell) per axis, for voxel_fast
    // storage element of logical voxel (x, y, z).  32-bit, and every product through the 24-bit multiplier (v_mad_u32_u24, full rate; a
    // 32 x 32 multiply is quarter rate on CDNA): kt_raycast_impl requires N <= 1536, so Z N + Y < N^2 < 2^24 and the result < N^3 < 2^32.
    __device__ __forceinline__ unsigned int index(int x, int y, int z) const
    {
        int X = x + a.wx; if (X >= a.N) X -= a.N;
        int Y = y + a.wy; if (Y >= a.N) Y -= a.N;
        int Z = z + a.wz; if (Z >= a.N) Z -= a.N;
        return kt_mad24(kt_mad24((unsigned int)Z, (unsigned int)a.N, (unsigned int)Y), (unsigned int)a.N, (unsigned int)X);
    }
    __device__ __forceinline__ void voxel(float px, float py, float pz, int& gx, int& gy, int& gz) const
    {
        gx = kt_f2i_rd(px / a.cx_);
        gy = kt_f2i_rd(py / a.cy_);
        gz = kt_f2i_rd(pz / a.cz_);
    }
    // getVoxel for the march loop.  floor(RN(p / cell)) is needed bit-exactly, but the quotient itself only matters next
    // to an integer: q' = p * RN(1 / cell) differs from RN(p / cell) by at most 3 * 2^-24 * |q| (< 1e-4 for |q| <= 512), so
    // whenever q' is farther than 2e-4 from an integer (and |q'| < 1024) floor(q') is the reference's voxel.  Lanes that are
    // not provably safe take the correctly rounded division; the branch is wave-uniform and rare (~7% of wave steps).
    __device__ __forceinline__ void voxel_fast(float px, float py, float pz, int& gx, int& gy, int& gz) const
    {
        float qx = px * rcx, qy = py * rcy, qz = pz * rcz;
        const float fx = qx - __builtin_floorf(qx), fy = qy - __builtin_floorf(qy), fz = qz - __builtin_floorf(qz);
        const float lo = 2e-4f, hi = 1.0f - 2e-4f;
        const bool safe = fx > lo && fx < hi && fy > lo && fy < hi && fz > lo && fz < hi &&
                          fabsf(qx) < 1024.f && fabsf(qy) < 1024.f && fabsf(qz) < 1024.f;
        if (!safe) { qx = px / a.cx_; qy = py / a.cy_; qz = pz / a.cz_; }
        gx = kt_f2i_rd(qx); gy = kt_f2i_rd(qy); gz = kt_f2i_rd(qz);
    }
    // interpolateTrilineary / ...Color / ...Heat bodies, ray_caster.cu:160-296, ONE AXIS AT A TIME (round 5).  Everything such a call derives
    // from a coordinate -- its voxel floor(p / cell), the in-volume test, the lower tap after the half-cell comparison, the fraction and the two
    // wrapped storage offsets -- depends on that coordinate alone, and a hit evaluates 12 interpolations at points that share coordinates: the
    // four colour channels sit at ONE point, and each of the six normal taps (ray_caster.cu:389-409) moves ONE coordinate of that point by a
    // cell.  15 axis evaluations instead of 36, each with the reference's expressions (the voxel through kt_rc::voxel_fast's rule, applied
    // per axis: floor(p * RN(1 / cell)) where provably equal to floor(RN(p / cell)), the division otherwise).
    struct axis_t {
        int gpre;            // floor(p / cell): getVoxel's coordinate
        bool ok;             // 0 < gpre < N - 1 (ray_caster.cu:166-173)
        float f;             // (p - (g + 0.5) cell) / cell for the lower tap g
        unsigned int o[2];   // storage offset contributions of taps g and g + 1 along this axis (wrapped; x: 1, y: N, z: N^2)
    };
    template <int AX>
    __device__ __forceinline__ axis_t axis(float p) const
    {
        const float cell = AX == 0 ? a.cx_ : AX == 1 ? a.cy_ : a.cz_, rcell = AX == 0 ? rcx : AX == 1 ? rcy : rcz;
        const int w = AX == 0 ? a.wx : AX == 1 ? a.wy : a.wz, N = a.N;
        float q = p * rcell;
        const float fr = q - __builtin_floorf(q);
        const bool safe = fr > 2e-4f && fr < 1.0f - 2e-4f && fabsf(q) < 1024.f;
        if (!safe) q = p / cell;   // (wave-uniform skip when every lane is safe)
        axis_t r;
        int g = kt_f2i_rd(q);
        r.gpre = g;
        r.ok = !(g <= 0 || g >= N - 1);
        const float v = ((float)g + 0.5f) * cell;
        g = (p < v) ? (g - 1) : g;
        // (a short form of this division by a launch constant -- reciprocal + two residual corrections, verified exhaustively per
        // divisor -- was measured in round 3: no difference in the kernel's time; profiles/r03_experiments.md)
        r.f = __builtin_fmaf(-((float)g + 0.5f), cell, p) / cell;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            int t = g + d + w; if (t >= N) t -= N;
            r.o[d] = AX == 0 ? (unsigned int)t : AX == 1 ? kt_mul24((unsigned int)t, (unsigned int)N) : kt_mul24(kt_mul24((unsigned int)t, (unsigned int)N), (unsigned int)N);
        }
        return r;
    }
    // the 8-tap blend of interpolateTrilineary (ray_caster.cu:185-194), taps r[k] with k = 4 dx + 2 dy + dz
    static __device__ __forceinline__ float blend(const float (&r)[8], float fa, float fb, float fc)
    {
        const float ia = 1 - fa, ib = 1 - fb, ic = 1 - fc;
        float res = __builtin_fmaf(r[0] * ia * ib, ic, r[1] * ia * ib * fc);
        res = __builtin_fmaf(r[2] * ia * fb, ic, res);
        res = __builtin_fmaf(r[3] * ia * fb, fc, res);
        res = __builtin_fmaf(r[4] * fa * ib, ic, res);
        res = __builtin_fmaf(r[5] * fa * ib, fc, res);
        res = __builtin_fmaf(r[6] * fa * fb, ic, res);
        res = __builtin_fmaf(r[7] * fa * fb, fc, res);
        return res;
    }
    // interpolateTrilineary at the point whose axes are (ax, ay, az); NaN where the reference returns "outside" (ray_caster.cu:166-173)
    __device__ __forceinline__ float tsdf_at(const axis_t& ax, const axis_t& ay, const axis_t& az) const
    {
        if (!(ax.ok && ay.ok && az.ok)) return kt_nan();
        float r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = kt_unpack_tsdf(a.volume[az.o[k & 1] + ay.o[(k >> 1) & 1] + ax.o[(k >> 2) & 1]]);
        return blend(r, ax.f, ay.f, az.f);
    }
    // interpolateTrilinearyColor for the four channels of one point (ray_caster.cu:372-378): each uchar4 tap is loaded once
    __device__ __forceinline__ uchar4 colour_at(const axis_t& ax, const axis_t& ay, const axis_t& az) const
    {
        if (!(ax.ok && ay.ok && az.ok)) return make_uchar4(0, 0, 0, 0);
        uchar4 c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = a.color[az.o[k & 1] + ay.o[(k >> 1) & 1] + ax.o[(k >> 2) & 1]];
        float r0[8], r1[8], r2[8], r3[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { r0[k] = (float)c[k].x; r1[k] = (float)c[k].y; r2[k] = (float)c[k].z; r3[k] = (float)c[k].w; }
        return make_uchar4(kt_f2u8_rz(blend(r0, ax.f, ay.f, az.f)), kt_f2u8_rz(blend(r1, ax.f, ay.f, az.f)), kt_f2u8_rz(blend(r2, ax.f, ay.f, az.f)),
                           kt_f2u8_rz(blend(r3, ax.f, ay.f, az.f)));
    }
};

// 2x2 box down-sampling of one map level held in LDS (resizeMapKernel<normalize>, maps.cu:225-277): x-plane NaN in any of
// the four inputs -> NaN in the output x plane only.  src: [3][S][S] tile in LDS, dst written to LDS tile and to global.
template <bool NORMALIZE, int S>
__device__ __forceinline__ void kt_tile_resize(const float* __restrict__ src, float* __restrict__ dst, int lx, int ly, int gx, int gy, int dcols,
                                               int drows, float* __restrict__ out)
{
    constexpr int D = S / 2;
    const float x00 = src[(2 * ly) * S + 2 * lx], x01 = src[(2 * ly) * S + 2 * lx + 1];
    const float x10 = src[(2 * ly + 1) * S + 2 * lx], x11 = src[(2 * ly + 1) * S + 2 * lx + 1];
    const bool inside = gx < dcols && gy < drows;
    if (kt_isnan(x00) || kt_isnan(x01) || kt_isnan(x10) || kt_isnan(x11)) {
        dst[ly * D + lx] = kt_nan();
        if (inside) out[gy * dcols + gx] = kt_nan();
        return;
    }
    f3 n;
    n.x = (x00 + x01 + x10 + x11) / 4;
    const float* sy = src + S * S;
    n.y = (sy[(2 * ly) * S + 2 * lx] + sy[(2 * ly) * S + 2 * lx + 1] + sy[(2 * ly + 1) * S + 2 * lx] + sy[(2 * ly + 1) * S + 2 * lx + 1]) / 4;
    const float* sz = src + 2 * S * S;
    n.z = (sz[(2 * ly) * S + 2 * lx] + sz[(2 * ly) * S + 2 * lx + 1] + sz[(2 * ly + 1) * S + 2 * lx] + sz[(2 * ly + 1) * S + 2 * lx + 1]) / 4;
    if (NORMALIZE) n = kt_normalized(n);
    dst[ly * D + lx] = n.x;
    dst[D * D + ly * D + lx] = n.y;
    dst[2 * D * D + ly * D + lx] = n.z;
    if (inside) {
        out[gy * dcols + gx] = n.x;
        out[(gy + drows) * dcols + gx] = n.y;
        out[(gy + 2 * drows) * dcols + gx] = n.z;
    }
}

#ifndef KT_RC_BATCH
#define KT_RC_BATCH 6   // samples issued together per march iteration (round 5 A/B: 2: 57.2 us, 4: 52.6, 6: 51.4, 8: 52.2 serial stage at VGA; 255 / 243 / 244 us at 1280x960 for 4 / 6 / 8)
#endif
#define KT_RC_MAX_BRICKS 32768   // brick flags staged in LDS (N <= 1024)

// Empty-space skipping (SKIP).  The march only ever reacts to a sign change between two consecutive samples (+ -> - is the hit,
// - -> + leaves).  tsdf23 keeps one flag per 32^3 storage brick, raised when a negative value is stored into it and never lowered
// (clears only zero voxels, so a stale flag is merely conservative).  A sample inside an unflagged brick is >= 0: together with a
// non-negative predecessor it can trigger neither exit, so the whole run of samples up to the brick's far face is replaced by
// the same number of `time_curr += time_step` float adds (the sample times are DEFINED by that recurrence) -- ~80 instructions
// per brick instead of ~100 per sample.  The hop is taken only when every live lane of the wave can take it; the brick test
// uses a 0.01-voxel margin so that position rounding cannot move a skipped sample across a face.  The value of the last
// skipped sample is fetched lazily when the next normal step needs it as `tsdf_prev`.
template <bool COUNT, bool PYR, bool SKIP>
__global__ __launch_bounds__(256) void kt_raycast_kernel(const kt_raycast_args a_in)
{
    kt_raycast_args a = a_in;
    if (a.fp) {
        if (a.fp->skip) return;  // parked for the host's shift path; the predicted maps are rebuilt there
#pragma unroll
        for (int k = 0; k < 9; ++k) a.R.m[k] = a.fp->R[k];
        a.tx = a.fp->t[0]; a.ty = a.fp->t[1]; a.tz = a.fp->t[2];
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char s_bricks[];
    if (SKIP) {
        const int nbytes = a.nb * a.nb * a.nb;  // a multiple of 4: nb^3 with N % 32 == 0 is not guaranteed to be, so copy bytes
        for (int i = threadIdx.x; i < nbytes; i += 256) s_bricks[i] = a.bricks[i];
        __syncthreads();
    }
    // a 256-thread block covers a 16x16 pixel tile; each wave an 8x8 sub-tile (coherent gathers)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = (wave & 1) * 8 + (lane & 7), ty = (wave >> 1) * 8 + (lane >> 3);
    const int x = blockIdx.x * 16 + tx;
    const int y = blockIdx.y * 16 + ty;
    const bool in_image = x < a.cols && y < a.rows;
    const kt_rc rc{a, 1.0f / a.cx_, 1.0f / a.cy_, 1.0f / a.cz_};
    const int cols = a.cols, rows = a.rows, N = a.N;
    unsigned int steps = 0, hopped = 0, n_hop_iters = 0, n_batch_iters = 0;

    float out_vx = kt_nan(), out_nx = kt_nan();
    bool hit = false, has_normal = false;
    float vfx = 0, vfy = 0, vfz = 0, nx = 0, ny = 0, nz = 0;
    uchar4 colr = make_uchar4(0, 0, 0, 0);

    if (in_image) {
        const f3 rs = {a.tx, a.ty, a.tz};
        f3 rnl = {((float)x - a.intr.cx) / a.intr.fx, ((float)y - a.intr.cy) / a.intr.fy, 1.0f};
        f3 ray_next = kt_add(kt_mul(a.R, rnl), rs);
        f3 rd = kt_normalized(kt_sub(ray_next, rs));
        rd.x = (rd.x == 0.f) ? (float)1e-15 : rd.x;
        rd.y = (rd.y == 0.f) ? (float)1e-15 : rd.y;
        rd.z = (rd.z == 0.f) ? (float)1e-15 : rd.z;
        // getMinTime / getMaxTime  ray_caster.cu:56-74
        const float txmin = ((rd.x > 0 ? 0.f : a.vsx) - rs.x) / rd.x;
        const float tymin = ((rd.y > 0 ? 0.f : a.vsy) - rs.y) / rd.y;
        const float tzmin = ((rd.z > 0 ? 0.f : a.vsz) - rs.z) / rd.z;
        const float txmax = ((rd.x > 0 ? a.vsx : 0.f) - rs.x) / rd.x;
        const float tymax = ((rd.y > 0 ? a.vsy : 0.f) - rs.y) / rd.y;
        const float tzmax = ((rd.z > 0 ? a.vsz : 0.f) - rs.z) / rd.z;
        float time_start_volume = fmaxf(fmaxf(txmin, tymin), tzmin);
        const float time_exit_volume = fminf(fminf(txmax, tymax), tzmax);
        time_start_volume = fmaxf(time_start_volume, 0.f);
        if (time_start_volume < time_exit_volume) {
            float time_curr = time_start_volume;
            int gx, gy, gz;
            rc.voxel(__builtin_fmaf(rd.x, time_curr, rs.x), __builtin_fmaf(rd.y, time_curr, rs.y), __builtin_fmaf(rd.z, time_curr, rs.z), gx, gy, gz);
            gx = max(0, min(gx, N - 1)); gy = max(0, min(gy, N - 1)); gz = max(0, min(gz, N - 1));
            // only the SIGN of the nearest-voxel tsdf steers the march: compare the packed shorts directly
            int tsdf = a.volume[rc.index(gx, gy, gz)];
            const float max_time = 3 * (a.vsx + a.vsy + a.vsz);
            const float rcx = rc.rcx, rcy = rc.rcy, rcz = rc.rcz;
            // The march (ray_caster.cu:340-352) visits time_curr, time_curr + step, ... one dependent gather per step.
            // The addresses do not depend on the loaded values, so KT_RC_BATCH steps are issued together and the exit
            // tests are then replayed in order; loads past the exit point are speculative and always in bounds.
            // The loop below is branch-free per step (selects instead of breaks) with 32-bit voxel offsets; the only
            // branches are the wave-uniform slow path of the voxel index and the batch-level exit.
            bool crossing = false, done = false;
            float t_cross = 0.f;
            const unsigned int uN = (unsigned int)N;
            // SKIP state: tsdf_known == false means "the previous sample was skipped: its value is >= 0 but not loaded"
            bool tsdf_known = true;
            const float fN = (float)N, eps_v = 0.01f;
            const float rwx = (float)(a.wx & (KT_BRICK - 1)), rwy = (float)(a.wy & (KT_BRICK - 1)), rwz = (float)(a.wz & (KT_BRICK - 1));
            const int awx = a.wx >> KT_BRICK_LOG2, awy = a.wy >> KT_BRICK_LOG2, awz = a.wz >> KT_BRICK_LOG2;
            const float fB = (float)KT_BRICK, rB = 1.0f / (float)KT_BRICK;
            // voxels per unit of ray time along each axis, and its inverse (geometry only: the margins absorb their rounding)
            const float vdx = rd.x * rcx, vdy = rd.y * rcy, vdz = rd.z * rcz;
            const float ivx = __builtin_amdgcn_rcpf(vdx), ivy = __builtin_amdgcn_rcpf(vdy), ivz = __builtin_amdgcn_rcpf(vdz);
            const float inv_step = __builtin_amdgcn_rcpf(a.time_step);
            while (true) {
                if (SKIP) {
                    const float tn = time_curr + a.time_step;
                    const float qx = __builtin_fmaf(rd.x, tn, rs.x) * rcx, qy = __builtin_fmaf(rd.y, tn, rs.y) * rcy,
                                qz = __builtin_fmaf(rd.z, tn, rs.z) * rcz;
                    // cell of the storage-brick grid seen from logical coordinates: faces at 32 c - (wrap & 31)
                    const float cx = __builtin_floorf((qx + rwx) * rB), cy = __builtin_floorf((qy + rwy) * rB),
                                cz = __builtin_floorf((qz + rwz) * rB);
                    const float lox = fmaxf(__builtin_fmaf(cx, fB, -rwx), 0.0f) + eps_v, hix = fminf(__builtin_fmaf(cx, fB, fB - rwx), fN) - eps_v;
                    const float loy = fmaxf(__builtin_fmaf(cy, fB, -rwy), 0.0f) + eps_v, hiy = fminf(__builtin_fmaf(cy, fB, fB - rwy), fN) - eps_v;
                    const float loz = fmaxf(__builtin_fmaf(cz, fB, -rwz), 0.0f) + eps_v, hiz = fminf(__builtin_fmaf(cz, fB, fB - rwz), fN) - eps_v;
                    const bool inside = qx > lox && qx < hix && qy > loy && qy < hiy && qz > loz && qz < hiz;
                    bool canhop = false;
                    float dt = 0.f;
                    if (!done && inside && (!tsdf_known || tsdf >= 0)) {
                        int bx = kt_cvt_i32(cx) + awx; bx -= (bx >= a.nb) ? a.nb : 0;
                        int by = kt_cvt_i32(cy) + awy; by -= (by >= a.nb) ? a.nb : 0;
                        int bz = kt_cvt_i32(cz) + awz; bz -= (bz >= a.nb) ? a.nb : 0;
                        if (s_bricks[kt_mad24(kt_mad24((unsigned int)bz, (unsigned int)a.nb, (unsigned int)by), (unsigned int)a.nb, (unsigned int)bx)] == 0) {
                            // ray time from this sample to the (margin-shrunk) far face of the cell
                            const float dx = ((vdx > 0 ? hix : lox) - qx) * ivx, dy = ((vdy > 0 ? hiy : loy) - qy) * ivy,
                                        dz = ((vdz > 0 ? hiz : loz) - qz) * ivz;
                            dt = fminf(fminf(dx, dy), dz);
                            canhop = dt >= 0.0f && tn + dt < max_time;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(!done && !canhop) == 0) {
                        // samples tn + j * step, j = 0 .. K - 1, lie inside the cell
                        int K = canhop ? 1 + kt_cvt_i32(dt * inv_step) : 0;
                        K = min(K, 64);   // a 32^3 brick holds at most ~14 samples; the bound keeps the replayed adds' drift << the margin
                        const int kmax = kt_wave_max(K);
                        for (int i = 0; i < kmax; ++i) time_curr = (i < K) ? time_curr + a.time_step : time_curr;
                        if (COUNT) { steps += (unsigned int)K; hopped += (unsigned int)K; if (lane == 0) ++n_hop_iters; }
                        tsdf_known = tsdf_known && K == 0;
                        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                        continue;
                    }
                    if (!tsdf_known && !done) {  // the next normal step needs the skipped predecessor's value
                        int gx2, gy2, gz2;
                        rc.voxel(__builtin_fmaf(rd.x, time_curr, rs.x), __builtin_fmaf(rd.y, time_curr, rs.y), __builtin_fmaf(rd.z, time_curr, rs.z),
                                 gx2, gy2, gz2);
                        tsdf = a.volume[rc.index(gx2, gy2, gz2)];
                    }
                    tsdf_known = true;
                }
                if (COUNT && lane == 0) ++n_batch_iters;
                float tc[KT_RC_BATCH];
                unsigned int gi[KT_RC_BATCH];
                bool inb[KT_RC_BATCH];
                float t = time_curr;
#pragma unroll
                for (int k = 0; k < KT_RC_BATCH; ++k) {
                    tc[k] = t;
                    const float tn = t + a.time_step;
                    const float px = __builtin_fmaf(rd.x, tn, rs.x), py = __builtin_fmaf(rd.y, tn, rs.y), pz = __builtin_fmaf(rd.z, tn, rs.z);
                    // getVoxel: floor(RN(p / cell)), via p * RN(1 / cell) when provably identical (see kt_rc::voxel_fast)
                    float qx = px * rcx, qy = py * rcy, qz = pz * rcz;
                    float flx = __builtin_floorf(qx), fly = __builtin_floorf(qy), flz = __builtin_floorf(qz);
                    // every fraction farther than 2e-4 from an integer, every |q| < 1024: two max3 and two compares (a NaN coordinate
                    // slips through the max, and floor(NaN) converts to voxel 0 on either path)
                    const float fx = qx - flx, fy = qy - fly, fz = qz - flz;
                    const float off = __builtin_fmaxf(__builtin_fmaxf(fabsf(fx - 0.5f), fabsf(fy - 0.5f)), fabsf(fz - 0.5f));
                    const float big = __builtin_fmaxf(__builtin_fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));
                    const bool safe = off < 0.5f - 2e-4f && big < 1024.f;
                    if (!safe) {   // (the compiler skips the block when no lane of the wave needs it)
                        flx = __builtin_floorf(px / a.cx_);
                        fly = __builtin_floorf(py / a.cy_);
                        flz = __builtin_floorf(pz / a.cz_);
                    }
                    const unsigned int ux = (unsigned int)kt_cvt_i32(flx), uy = (unsigned int)kt_cvt_i32(fly), uz = (unsigned int)kt_cvt_i32(flz);
                    inb[k] = ux < uN && uy < uN && uz < uN;  // checkInds (negative indices wrap to huge unsigned values)
                    unsigned int X = ux + (unsigned int)a.wx; X -= (X >= uN) ? uN : 0u;
                    unsigned int Y = uy + (unsigned int)a.wy; Y -= (Y >= uN) ? uN : 0u;
                    unsigned int Z = uz + (unsigned int)a.wz; Z -= (Z >= uN) ? uN : 0u;
                    const unsigned int gidx = kt_mad24u(kt_mad24u(Z, uN, Y), uN, X);   // (garbage outside the volume: masked)
                    gi[k] = inb[k] ? gidx : 0u;
                    t += a.time_step;
                }
                short v[KT_RC_BATCH];
#pragma unroll
                for (int k = 0; k < KT_RC_BATCH; ++k) v[k] = a.volume[gi[k]];
#pragma unroll
                for (int k = 0; k < KT_RC_BATCH; ++k) {
                    // for (; time_curr < max_time; ...) { if (!checkInds) break; ... }  replayed with selects
                    const bool alive = !done && (tc[k] < max_time) && inb[k];
                    const int cur = v[k];
                    const bool mp = alive && tsdf < 0 && cur > 0;   // - -> + : leave without a hit
                    const bool pm = alive && tsdf > 0 && cur < 0;   // + -> - : zero crossing
                    if (COUNT) steps += alive ? 1u : 0u;
                    t_cross = pm ? tc[k] : t_cross;
                    crossing = crossing || pm;
                    tsdf = alive ? cur : tsdf;
                    done = done || !alive || mp || pm;
                }
                time_curr = t;
                if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
            }
            if (crossing) {  // zero crossing, ray_caster.cu:354-422
                time_curr = t_cross;
                const float tn = time_curr + a.time_step;
                const float px = __builtin_fmaf(rd.x, tn, rs.x), py = __builtin_fmaf(rd.y, tn, rs.y), pz = __builtin_fmaf(rd.z, tn, rs.z);
                const float Ftdt = rc.tsdf_at(rc.axis<0>(px), rc.axis<1>(py), rc.axis<2>(pz));
                if (!kt_isnan(Ftdt)) {
                    const float qx = __builtin_fmaf(rd.x, time_curr, rs.x), qy = __builtin_fmaf(rd.y, time_curr, rs.y), qz = __builtin_fmaf(rd.z, time_curr, rs.z);
                    const kt_rc::axis_t qax = rc.axis<0>(qx), qay = rc.axis<1>(qy), qaz = rc.axis<2>(qz);
                    const float Ft = rc.tsdf_at(qax, qay, qaz);
                    if (!kt_isnan(Ft)) {
                        const float Ts = time_curr - a.time_step * Ft / (Ftdt - Ft);
                        vfx = __builtin_fmaf(rd.x, Ts, rs.x); vfy = __builtin_fmaf(rd.y, Ts, rs.y); vfz = __builtin_fmaf(rd.z, Ts, rs.z);
                        hit = true;
                        out_vx = vfx;
                        const int hx = qax.gpre, hy = qay.gpre, hz = qaz.gpre;   // getVoxel(ray_start + ray_dir * time_curr), ray_caster.cu:380
                        const kt_rc::axis_t cax = rc.axis<0>(vfx), cay = rc.axis<1>(vfy), caz = rc.axis<2>(vfz);
                        colr = rc.colour_at(cax, cay, caz);
                        if (hx > 1 && hy > 1 && hz > 1 && hx < N - 2 && hy < N - 2 && hz < N - 2) {
                            const float Fx1 = rc.tsdf_at(rc.axis<0>(vfx + a.cx_), cay, caz), Fx2 = rc.tsdf_at(rc.axis<0>(vfx - a.cx_), cay, caz);
                            const float Fy1 = rc.tsdf_at(cax, rc.axis<1>(vfy + a.cy_), caz), Fy2 = rc.tsdf_at(cax, rc.axis<1>(vfy - a.cy_), caz);
                            const float Fz1 = rc.tsdf_at(cax, cay, rc.axis<2>(vfz + a.cz_)), Fz2 = rc.tsdf_at(cax, cay, rc.axis<2>(vfz - a.cz_));
                            const f3 n = kt_normalized({Fx1 - Fx2, Fy1 - Fy2, Fz1 - Fz2});
                            nx = n.x; ny = n.y; nz = n.z;
                            has_normal = true;
                            out_nx = nx;
                        }
                    }
                }
            }
        }
        // unhit pixels: NaN in the x planes only, y/z planes and the colour map keep their previous content
        a.vmap[y * cols + x] = out_vx;
        a.nmap[y * cols + x] = out_nx;
        if (hit) {
            a.vmap[(y + rows) * cols + x] = vfy;
            a.vmap[(y + 2 * rows) * cols + x] = vfz;
            a.vmap_color[y * cols + x] = colr;
            if (has_normal) {
                a.nmap[(y + rows) * cols + x] = ny;
                a.nmap[(y + 2 * rows) * cols + x] = nz;
            }
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { steps += __shfl_down(steps, off, 64); hopped += __shfl_down(hopped, off, 64); }
        if (lane == 0 && steps) atomicAdd(a.steps, (unsigned long long)steps);
        if (lane == 0 && hopped) atomicAdd(a.steps + 1, (unsigned long long)hopped);  // diagnostics: samples replaced by brick hops
        if (lane == 0) { atomicAdd(a.steps + 2, (unsigned long long)n_hop_iters); atomicAdd(a.steps + 3, (unsigned long long)n_batch_iters); }
    }
    if (PYR) {
        // fused resizeVMap / resizeNMap for levels 1..3 of this 16x16 tile (KintinuousTracker.cpp:892-899): 8x8, 4x4, 2x2
        __shared__ float tv0[3 * 16 * 16], tn0[3 * 16 * 16];
        __shared__ float tv1[3 * 8 * 8], tn1[3 * 8 * 8], tv2[3 * 4 * 4], tn2[3 * 4 * 4], tv3[3 * 2 * 2], tn3[3 * 2 * 2];
        const int li = ty * 16 + tx;
        tv0[li] = out_vx; tv0[256 + li] = vfy; tv0[512 + li] = vfz;
        tn0[li] = out_nx; tn0[256 + li] = ny; tn0[512 + li] = nz;
        __syncthreads();
        const int t = threadIdx.x;
        if (t < 64) {
            const int lx = t & 7, ly = t >> 3;
            kt_tile_resize<false, 16>(tv0, tv1, lx, ly, blockIdx.x * 8 + lx, blockIdx.y * 8 + ly, cols / 2, rows / 2, a.vpyr[0]);
        } else if (t < 128) {
            const int lx = t & 7, ly = (t >> 3) & 7;
            kt_tile_resize<true, 16>(tn0, tn1, lx, ly, blockIdx.x * 8 + lx, blockIdx.y * 8 + ly, cols / 2, rows / 2, a.npyr[0]);
        }
        __syncthreads();
        if (t < 16) {
            const int lx = t & 3, ly = t >> 2;
            kt_tile_resize<false, 8>(tv1, tv2, lx, ly, blockIdx.x * 4 + lx, blockIdx.y * 4 + ly, cols / 4, rows / 4, a.vpyr[1]);
        } else if (t >= 64 && t < 80) {
            const int lx = t & 3, ly = (t >> 2) & 3;
            kt_tile_resize<true, 8>(tn1, tn2, lx, ly, blockIdx.x * 4 + lx, blockIdx.y * 4 + ly, cols / 4, rows / 4, a.npyr[1]);
        }
        __syncthreads();
        if (t < 4) {
            const int lx = t & 1, ly = t >> 1;
            kt_tile_resize<false, 4>(tv2, tv3, lx, ly, blockIdx.x * 2 + lx, blockIdx.y * 2 + ly, cols / 8, rows / 8, a.vpyr[2]);
        } else if (t >= 64 && t < 68) {
            const int lx = t & 1, ly = (t >> 1) & 1;
            kt_tile_resize<true, 4>(tn2, tn3, lx, ly, blockIdx.x * 2 + lx, blockIdx.y * 2 + ly, cols / 8, rows / 8, a.npyr[2]);
        }
    }
}

int kt_raycast_impl(kt_ctx* c, const kt_intr* intr, const kt_mat33* Rcurr, const float tcurr[3], float tranc_dist,
                    const float volume_size[3], const int16_t* volume, float* vmap, float* nmap, int cols, int rows,
                    const int voxel_wrap[3], uint8_t* vmap_curr_color, const uint8_t* color_volume, int N,
                    unsigned long long* steps_dev, float* const* vpyr, float* const* npyr, const kt_frame_params* fp,
                    const unsigned char* bricks)
{
    KT_ARG(c && intr && Rcurr && tcurr && volume_size && volume && vmap && nmap && voxel_wrap && vmap_curr_color && color_volume);
    KT_ARG(N > 0 && cols > 0 && rows > 0);
    KT_ARG(N <= 1536);  // 32-bit voxel offsets (N^3 < 2^32)
    kt_raycast_args a;
    a.R = *Rcurr;
    a.tx = tcurr[0]; a.ty = tcurr[1]; a.tz = tcurr[2];
    a.time_step = tranc_dist * 0.8f;  // ray_caster.cu:444
    a.vsx = volume_size[0]; a.vsy = volume_size[1]; a.vsz = volume_size[2];
    a.cx_ = volume_size[0] / N; a.cy_ = volume_size[1] / N; a.cz_ = volume_size[2] / N;
    a.cols = cols; a.rows = rows; a.N = N;
    a.volume = volume;
    a.color = (const uchar4*)color_volume;
    a.intr = *intr;
    a.vmap = vmap; a.nmap = nmap;
    for (int k = 0; k < 3; ++k) KT_ARG(voxel_wrap[k] >= 0);
    a.wx = voxel_wrap[0] % N; a.wy = voxel_wrap[1] % N; a.wz = voxel_wrap[2] % N;
    a.vmap_color = (uchar4*)vmap_curr_color;
    a.steps = steps_dev;
    a.fp = fp;
    const bool pyr = vpyr && npyr;
    for (int k = 0; k < 3; ++k) { a.vpyr[k] = pyr ? vpyr[k] : nullptr; a.npyr[k] = pyr ? npyr[k] : nullptr; }
    if (pyr) KT_ARG((cols % 8) == 0 && (rows % 8) == 0);
    dim3 b(256), g(kt_div_up(cols, 16), kt_div_up(rows, 16));
    a.nb = N / KT_BRICK;
    const bool skip = bricks && (N % KT_BRICK) == 0 && a.nb * a.nb * a.nb <= KT_RC_MAX_BRICKS;
    a.bricks = skip ? bricks : nullptr;
    const size_t lds = skip ? (size_t)((a.nb * a.nb * a.nb + 15) & ~15) : 0;
#define KT_RC_LAUNCH(C, P, S) hipLaunchKernelGGL((kt_raycast_kernel<C, P, S>), g, b, lds, c->stream, a)
    if (skip) {
        if (pyr) { if (steps_dev) KT_RC_LAUNCH(true, true, true); else KT_RC_LAUNCH(false, true, true); }
        else { if (steps_dev) KT_RC_LAUNCH(true, false, true); else KT_RC_LAUNCH(false, false, true); }
    } else {
        if (pyr) { if (steps_dev) KT_RC_LAUNCH(true, true, false); else KT_RC_LAUNCH(false, true, false); }
        else { if (steps_dev) KT_RC_LAUNCH(true, false, false); else KT_RC_LAUNCH(false, false, false); }
    }
#undef KT_RC_LAUNCH
    KT_LAUNCH_CHECK();
    return KT_OK;
}

extern "C" int kt_raycast(kt_ctx* c, const kt_intr* intr, const kt_mat33* Rcurr, const float tcurr[3], float tranc_dist,
                          const float volume_size[3], const int16_t* volume, float* vmap, float* nmap, int cols, int rows,
                          const int voxel_wrap[3], uint8_t* vmap_curr_color, const uint8_t* color_volume, int N)
{
    return kt_raycast_impl(c, intr, Rcurr, tcurr, tranc_dist, volume_size, volume, vmap, nmap, cols, rows, voxel_wrap,
                           vmap_curr_color, color_volume, N, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// ================================================================================================
// a14  clearVolume{X,Y,Z}{,Back}{,c}                  tsdf_volume.cu:88-448
// The reference launches one thread per (x,y) [or (x,z)] walking the slab; here every wave clears
// 64 consecutive storage x of one line, 16 bytes per lane where the slab is contiguous in x.
// The set of cleared voxels reproduces the reference's launch geometry (incl. quirk A.15).
// ================================================================================================
struct kt_clear_args {
    void* vol; int elem_size; int N;
    int axis;     // 0 x, 1 y, 2 z
    int start;    // first storage index along the axis
    int count;    // number of planes (walk length), consecutive modulo N
    int xthreads; // X variants: number of x "threads" the reference launched (multiple of 16)
    int xw_log2;  // X variants: a wave covers 2^xw_log2 slab indices of 64 >> xw_log2 consecutive y rows
};

template <typename T>
__global__ __launch_bounds__(256) void kt_clear_kernel(const kt_clear_args a)
{
    const int N = a.N;
    T* vol = (T*)a.vol;
    if (a.axis == 0) {
        // grid: (ceil(count / xw), ceil(N / (4 * rows per wave)), N): x index inside the slab, y, z.  An X slab is a few voxels thin in the
        // fastest dimension: a wave takes xw = 2^xw_log2 >= slab width indices of 64 / xw rows, so that most of its lanes store.
        const int xw = 1 << a.xw_log2, lane = threadIdx.x & 63;
        const int i = blockIdx.x * xw + (lane & (xw - 1));
        const int y = (blockIdx.y * 4 + (threadIdx.x >> 6)) * (64 >> a.xw_log2) + (lane >> a.xw_log2);
        const int z = blockIdx.z;
        if (i >= a.count || i >= a.xthreads || y >= N) return;
        int x = a.start + i; if (x >= N) x -= N;
        vol[(size_t)x + (size_t)y * N + (size_t)z * N * N] = T(0);
    } else {
        // grid: (ceil(N/64), ceil(N/4), count): storage x, the free axis, plane inside the slab
        const int x = blockIdx.x * 64 + (threadIdx.x & 63);
        const int o = blockIdx.y * 4 + (threadIdx.x >> 6);
        const int p = blockIdx.z;
        if (x >= N || o >= N) return;
        int s = a.start + p; if (s >= N) s -= N;
        const size_t idx = a.axis == 1 ? (size_t)x + (size_t)s * N + (size_t)o * N * N
                                        : (size_t)x + (size_t)o * N + (size_t)s * N * N;
        vol[idx] = T(0);
    }
}

static int kt_wrap_base(int currentVoxelWrap, int N)
{
    return currentVoxelWrap > 0 ? currentVoxelWrap % N : N - ((-currentVoxelWrap) % N);
}

extern "C" int kt_clear_volume(kt_ctx* c, void* volume, int elem_size, int N, int axis, int back, int currentVoxelWrap,
                               int deltaVoxelWrap)
{
    KT_ARG(c && volume && (elem_size == 2 || elem_size == 4) && N > 0 && axis >= 0 && axis <= 2);
    kt_clear_args a;
    a.vol = volume; a.elem_size = elem_size; a.N = N; a.axis = axis; a.xthreads = N;
    const int base = kt_wrap_base(currentVoxelWrap, N);
    if (!back) {
        // clearVolumeIn{Y,Z}: walk bottom, bottom+1, ... for numUp+1 planes  (tsdf_volume.cu:240-262, 345-367)
        const int numUp = -(currentVoxelWrap - deltaVoxelWrap);
        a.start = base % N;
        a.count = numUp + 1;
    } else {
        // clearVolumeIn{Y,Z}Back: walk top, top-1, ... for numDown+1 planes   (tsdf_volume.cu:290-317, 395-422)
        const int top = (base + N) % N;
        const int numDown = currentVoxelWrap - deltaVoxelWrap;
        a.count = numDown + 1;
        int bottom = (top - numDown) % N;
        if (bottom < 0) bottom += N;
        a.start = bottom;
    }
    if (a.count <= 0) return KT_OK;
    if (a.count > N) a.count = N;
    if (axis == 0) {
        // clearVolumeX / XBack launch geometry (tsdf_volume.cu:117-237): only ceil(r/16)*16 x-threads exist
        int remainder = (deltaVoxelWrap - currentVoxelWrap) % 16;
        if (remainder != 0) remainder = (deltaVoxelWrap - currentVoxelWrap) + 16 - remainder;
        else remainder = abs(deltaVoxelWrap - currentVoxelWrap);
        int grid_x = (remainder + 15) / 16;
        if (grid_x <= 0) return KT_OK;
        a.xthreads = grid_x * 16;
    }
    dim3 b(256), g;
    a.xw_log2 = 6;
    if (axis == 0) {
        const int w = a.count < a.xthreads ? a.count : a.xthreads;   // indices that are actually written
        a.xw_log2 = 0;
        while ((1 << a.xw_log2) < w && a.xw_log2 < 6) ++a.xw_log2;
        g = dim3(kt_div_up(w, 1 << a.xw_log2), kt_div_up(N, 4 * (64 >> a.xw_log2)), N);
    }
    else g = dim3(kt_div_up(N, 64), kt_div_up(N, 4), a.count);
    if (elem_size == 2) hipLaunchKernelGGL(kt_clear_kernel<int16_t>, g, b, 0, c->stream, a);
    else hipLaunchKernelGGL(kt_clear_kernel<uint32_t>, g, b, 0, c->stream, a);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// ================================================================================================
// a15  extractCloudSlice -> extractKernelSlice        extract.cu:79-419
// One thread per candidate voxel of the slab (x fastest); wave-level compaction with a 64-lane ballot
// prefix and ONE global atomic per wave (the reference: 32-lane shared-memory scan + atomic per warp
// per z-step).  Output order is unspecified in both.
// ================================================================================================
struct kt_extract_args {
    const int16_t* volume; const uchar4* color;
    kt_point_xyzrgb* out; unsigned int out_cap;
    unsigned int* count;      // device counter
    float cx, cy, cz;         // cell size
    int wx, wy, wz;           // storage wrap (normalised)
    int rwx, rwy, rwz;        // real voxel wrap (for the output offset)
    int minX, maxX, minY, maxY, minZ, maxZ, subsample;
    int nx, ny, nz;           // extents of the scanned box
    int N;
};

__device__ __forceinline__ size_t kt_ex_index(const kt_extract_args& a, int x, int y, int z)
{
    // (x + wrap) % N with z + 1 possibly == N (wraps through the modulo, quirk of extract.cu:190-196)
    int X = (x + a.wx) % a.N, Y = (y + a.wy) % a.N, Z = (z + a.wz) % a.N;
    return (size_t)X + (size_t)Y * a.N + (size_t)Z * a.N * a.N;
}

__global__ __launch_bounds__(256) void kt_extract_kernel(const kt_extract_args a)
{
    const long long total = (long long)a.nx * a.ny * a.nz;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // (up to three points per thread, one per axis.  Indexed by the AXIS, a compile-time constant in the unrolled loops, never by a
    // running count: a dynamically indexed array lives in scratch memory, and the first launch of a scratch-using kernel on a queue makes
    // the runtime allocate it -- 1 ms in the middle of the first shift frame of a process.)
    float4 pts[3];
    uint32_t cols[3];
    bool found[3] = {false, false, false};
    if (tid < total) {
        const int ix = (int)(tid % a.nx);
        const int iy = (int)((tid / a.nx) % a.ny);
        const int iz = (int)(tid / ((long long)a.nx * a.ny));
        const int x = a.minX + ix, y = a.minY + iy, z = a.minZ + iz * a.subsample;
        const int N = a.N;
        if (x >= 0 && y >= 0 && x < N && y < N && x % a.subsample == 0 && y % a.subsample == 0) {
            const size_t i0 = kt_ex_index(a, x, y, z);
            const int W = a.color[i0].w;
            const float F = kt_unpack_tsdf(a.volume[i0]);
            if (W != 0 && F != 1.f) {
                const float V[3] = {((float)x + 0.5f) * a.cx, ((float)y + 0.5f) * a.cy, ((float)z + 0.5f) * a.cz};
                const float cell[3] = {a.cx, a.cy, a.cz};
#pragma unroll
                for (int axis = 0; axis < 3; ++axis) {
                    const int nx = x + (axis == 0), ny = y + (axis == 1), nz = z + (axis == 2);
                    if (axis == 0 && !(x + 1 < N)) continue;
                    if (axis == 1 && !(y + 1 < N)) continue;
                    const size_t in = kt_ex_index(a, nx, ny, nz);
                    const uchar4 cn = a.color[in];
                    const float Fn = kt_unpack_tsdf(a.volume[in]);
                    if (!(cn.w != 0 && Fn != 1.f)) continue;
                    if (!((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) continue;
                    float p[3] = {V[0], V[1], V[2]};
                    const float Vn = V[axis] + cell[axis];
                    const float d_inv = 1.f / (fabsf(F) + fabsf(Fn));
                    // dz: the other product is the contracted one (LLVM operand order on extract.cu:224, pinned by oracle/_ref)
                    p[axis] = (axis == 2 ? __builtin_fmaf(Vn, fabsf(F), V[axis] * fabsf(Fn)) : __builtin_fmaf(V[axis], fabsf(Fn), Vn * fabsf(F))) * d_inv;
                    pts[axis] = make_float4(p[0], p[1], p[2], 0.f);
                    // colour of the NEIGHBOUR voxel, alpha = weight of the base voxel; the point's byte order is
                    // b,g,r,a with ptr->b = colour.x and ptr->r = colour.z (store_point_type, quirk A.14)
                    cols[axis] = (uint32_t)cn.x | ((uint32_t)cn.y << 8) | ((uint32_t)cn.z << 16) | ((uint32_t)W << 24);
                    found[axis] = true;
                }
            }
        }
    }
    // wave compaction: exclusive prefix of `local` over the 64 lanes
    const int lane = threadIdx.x & 63;
    const int local = (int)found[0] + (int)found[1] + (int)found[2];
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const int wave_total = __shfl(incl, 63, 64);
    if (wave_total == 0) return;
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(a.count, (unsigned int)wave_total);
    base = __shfl(base, 0, 64);
    unsigned int o = base + (unsigned int)(incl - local);   // the thread's points in axis order, as before
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (!found[l]) continue;
        if (o < a.out_cap) {
            // store_point_type  extract.cu:307-317
            float4 lo, hi;
            // store_point_type extract.cu:307-317: the product realVoxelWrap * cell_size is loop invariant in the reference and ends up in
            // front of its loops, out of reach of the addition: not contracted (oracle/_ref pins it)
            lo.x = (pts[l].x + (float)a.rwx * a.cx) - ((a.cx * a.N) / 2);
            lo.y = (pts[l].y + (float)a.rwy * a.cy) - ((a.cy * a.N) / 2);
            lo.z = (pts[l].z + (float)a.rwz * a.cz) - ((a.cz * a.N) / 2);
            lo.w = 0.f;
            hi.x = __uint_as_float(cols[l]); hi.y = 0.f; hi.z = 0.f; hi.w = 0.f;
            float4* dst = (float4*)&a.out[o];  // two 16-byte stores per 32-byte point
            dst[0] = lo;
            dst[1] = hi;
        }
        ++o;
    }
}

int kt_extract_cloud_slice_async(kt_ctx* c, const int16_t* volume, const float volume_size[3], kt_point_xyzrgb* output,
                                 size_t output_capacity, const int voxel_wrap[3], const uint8_t* color_volume, int minX, int maxX,
                                 int minY, int maxY, int minZ, int maxZ, int subsample, const int real_voxel_wrap[3], int N,
                                 unsigned int* count_dev)
{
    KT_ARG(c && volume && volume_size && output && voxel_wrap && color_volume && real_voxel_wrap && count_dev);
    KT_ARG(N > 0 && subsample > 0);
    for (int k = 0; k < 3; ++k) KT_ARG(voxel_wrap[k] >= 0);
    KT_HIP(hipMemsetAsync(count_dev, 0, sizeof(unsigned int), c->stream));
    kt_extract_args a;
    a.volume = volume; a.color = (const uchar4*)color_volume;
    a.out = output; a.out_cap = (unsigned int)(output_capacity > 0xffffffffull ? 0xffffffffull : output_capacity);
    a.count = count_dev;
    a.cx = volume_size[0] / N; a.cy = volume_size[1] / N; a.cz = volume_size[2] / N;
    a.wx = voxel_wrap[0] % N; a.wy = voxel_wrap[1] % N; a.wz = voxel_wrap[2] % N;
    a.rwx = real_voxel_wrap[0]; a.rwy = real_voxel_wrap[1]; a.rwz = real_voxel_wrap[2];
    a.minX = minX < 0 ? 0 : minX; a.maxX = maxX > N ? N : maxX;
    a.minY = minY < 0 ? 0 : minY; a.maxY = maxY > N ? N : maxY;
    a.minZ = minZ; a.maxZ = maxZ; a.subsample = subsample;
    a.nx = a.maxX - a.minX; a.ny = a.maxY - a.minY;
    a.nz = (maxZ - minZ + subsample - 1) / subsample;  // for (z = minZ; z < maxZ; z += subsample)
    a.N = N;
    if (a.nx <= 0 || a.ny <= 0 || a.nz <= 0) return KT_OK;
    const long long total = (long long)a.nx * a.ny * a.nz;
    const long long blocks = (total + 255) / 256;
    KT_ARG(blocks < 0x7fffffffll);
    hipLaunchKernelGGL(kt_extract_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, a);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

extern "C" int kt_extract_cloud_slice(kt_ctx* c, const int16_t* volume, const float volume_size[3], kt_point_xyzrgb* output,
                                      size_t output_capacity, const int voxel_wrap[3], const uint8_t* color_volume, int minX,
                                      int maxX, int minY, int maxY, int minZ, int maxZ, int subsample,
                                      const int real_voxel_wrap[3], int N, size_t* count_host)
{
    KT_ARG(count_host);
    int s = kt_extract_cloud_slice_async(c, volume, volume_size, output, output_capacity, voxel_wrap, color_volume, minX, maxX,
                                         minY, maxY, minZ, maxZ, subsample, real_voxel_wrap, N, &c->counters[1]);
    if (s != KT_OK) return s;
    KT_HIP(hipMemcpyAsync(c->int_out_host, &c->counters[1], sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    size_t n = (size_t)(unsigned int)c->int_out_host[0];
    *count_host = n < output_capacity ? n : output_capacity;  // output_count = min(output.size, global_count)
    return KT_OK;
}
