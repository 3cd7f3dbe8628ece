/*
 * kt_oracle.h -- CPU restatement ("oracle") of the Kintinuous per-frame tracking + fusion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (kintinuous_amd/, include/, the C-ABI
 * library) may include, link or call this.  It is used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, and only as the checker / the CPU baseline.
 *
 * PARITY PINNED for the device kernels (rows a1-a15 + generateImage/generateDepth): the reference ships
 * no tests, golden vectors or fixtures for this path, but its nine .cu files compile for the host from
 * where they lie under /root/reference (oracle/Makefile target _ref/libkt_ref.so: a CUDA execution-model
 * shim, oracle/ref_shim/, runs every CUDA thread as a fiber with __syncthreads / __shfl_down / __ballot /
 * __all rendez-vous, FTZ like --ftz=true, fmad contraction on) and tests/test_oracle_vs_ref.py demands
 * that this restatement equals that build bit for bit on every output (one documented exception: the
 * bilateral filter's __expf, an approximate SFU function on the GPU -- glibc expf in the _ref build,
 * kto_expf here; the rounded u16 results differ by at most 1 in fewer than 1 pixel in 50 000 and the test
 * bounds exactly that).  tests/golden/golden_ref_v1.npz holds outputs of the _ref build for the GPU box.
 * The HOST logic (LDLT, Rodrigues, pose update, shift state machine: kt_oracle_host.c) restates Eigen 3.2 /
 * OpenCV 2.4 / PCL 1.7, none of which is vendored or installed: that part is pinned by analytic
 * known-answer tests only (tests/test_oracle_kat.py) -- "parity unpinned" for those functions.
 * Every function cites the reference file:line it follows (paths relative to /root/reference/src/).
 *
 * Arithmetic conventions (the irreducible gap to the nvcc build of the reference, which uses
 * --ftz=true --prec-div=false --prec-sqrt=false, CMakeLists.txt:47):
 *   - IEEE-754 binary32, round-to-nearest-even, denormals kept; '/' and sqrtf correctly rounded.
 *   - rsqrtf(x) is restated as 1.0f / sqrtf(x);  __expf is restated as kto_expf() (Cody-Waite +
 *     degree-6 polynomial in explicit fmaf steps, flushes below 2^-125).
 *   - compiled with -ffp-contract=off; a*b+c is fused ONLY where fmaf() is written explicitly.  The
 *     explicit fmaf() sites follow LLVM's contraction rule (fadd(fmul(a,b),c) -> fma(a,b,c), left
 *     operand first), which is what nvcc -fmad=true does to the reference source.
 *   - __float2int_rn/rz/rd saturate and map NaN to 0 (kto_f2i_*).
 *
 * Data layout: every image / map / volume is dense (row pitch == cols * sizeof(T)); the reference's
 * volume kernels already assume this (tsdf_volume.cu:612).
 *   vmap / nmap : float[3*rows][cols], planes x,y,z stacked by rows; invalid = NaN in the x plane.
 *   volume      : short[N*N*N], index x + y*N + z*N*N, wrapped storage ((x+wx)%N ...).
 *   colour vol  : uint8[N*N*N][4] = r,g,b,weight.
 */
#ifndef KT_ORACLE_H_
#define KT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float fx, fy, cx, cy; } kto_intr;          /* internal.h:249-260 (Intr)      */
typedef struct { float m[9]; } kto_mat33;                   /* internal.h:279-282 (Mat33, row-major) */
typedef struct { float v[29]; } kto_jtj;                    /* internal.h:98-149 (JtJJtrSE3)  */
typedef struct { int16_t zero_x, zero_y, one_x, one_y; float diff; uint8_t valid; uint8_t pad[3]; } kto_dataterm; /* internal.h:90-96 */
typedef struct { float x, y, z, pad0; uint8_t b, g, r, a; uint32_t pad1[3]; } kto_point; /* internal.h:156-184, 32 B */

/* ---- scalar helpers (exported so tests can probe the conventions) ---- */
int   kto_f2i_rn(float x);
int   kto_f2i_rz(float x);
int   kto_f2i_rd(float x);
float kto_expf(float x);

/* ---- a1-a5, a13: image-side kernels ---- */
void kto_bilateral_filter(const uint16_t* src, uint16_t* dst, int cols, int rows);
void kto_pyr_down(const uint16_t* src, int scols, int srows, uint16_t* dst);
void kto_create_vmap(kto_intr intr, const uint16_t* depth, int cols, int rows, float* vmap);
void kto_create_nmap(const float* vmap, int cols, int rows, float* nmap);
void kto_transform_maps(const float* vmap_src, const float* nmap_src, int cols, int rows,
                        const kto_mat33* R, const float t[3], float* vmap_dst, float* nmap_dst);
void kto_resize_vmap(const float* in, int in_cols, int in_rows, float* out);
void kto_resize_nmap(const float* in, int in_cols, int in_rows, float* out);
/* generateImage / generateDepth (image_generator.cu): dst, dst_color = rgb24 [rows][cols][3]; vmap_curr_color = uchar4 per pixel */
void kto_generate_image(const float* vmap, const float* nmap, const uint8_t* vmap_curr_color, int cols, int rows,
                        const float light_pos[3], int light_number, uint8_t* dst, uint8_t* dst_color);
void kto_generate_depth(const kto_mat33* R_inv, const float t[3], const float* vmap, const float* nmap, int cols, int rows, uint16_t* dst);

/* ---- a10 helpers: RGB-D image pyramids ---- */
void kto_depth_to_metres(const uint16_t* src, float* dst, int cols, int rows, int cutoff);
void kto_bgr_to_intensity(const uint8_t* src_rgb24, uint8_t* dst, int cols, int rows);
void kto_pyr_down_gauss_f32(const float* src, int scols, int srows, float* dst);
void kto_pyr_down_gauss_u8(const uint8_t* src, int scols, int srows, uint8_t* dst);
void kto_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy);
void kto_project_to_cloud(const float* depth, int cols, int rows, float* cloud_xyz,
                          double fx, double fy, double cx, double cy, int level);

/* ---- a6, a8, a9: tracking reductions.  order: 0 = reference float tree, 1 = double accumulation ---- */
void kto_icp_step(const kto_mat33* Rcurr, const float tcurr[3], const float* vmap_curr, const float* nmap_curr,
                  const kto_mat33* Rprev_inv, const float tprev[3], kto_intr intr,
                  const float* vmap_g_prev, const float* nmap_g_prev, int cols, int rows,
                  float dist_thres, float angle_thres, int order,
                  float A[36], float b[6], float residual[2]);
void kto_rgb_residual(float min_scale, const int16_t* dIdx, const int16_t* dIdy,
                      const float* last_depth, const float* next_depth,
                      const uint8_t* last_image, const uint8_t* next_image, int cols, int rows,
                      kto_dataterm* corres, float max_depth_delta, const float kt[3], const kto_mat33* krkinv,
                      int* sigma_sum, int* count);
void kto_rgb_step(const kto_dataterm* corres, float sigma, const float* cloud_xyz, float fx, float fy,
                  const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows, int order,
                  float A[36], float b[6]);

/* ---- a11, a12, a14, a15: volume kernels ---- */
void kto_init_volume(int16_t* vol, int N);
void kto_init_color_volume(uint8_t* cvol, int N);
void kto_scale_depth(const uint16_t* depth, float* scaled, int cols, int rows, kto_intr intr, int angle_color);
/* returns U = number of voxels that pass the update predicate */
long long kto_integrate_tsdf(const uint16_t* depth_raw, int cols, int rows, kto_intr intr, const float volume_size[3],
                             const kto_mat33* Rcurr_inv, const float tcurr[3], float tranc_dist,
                             int16_t* volume, float* depth_scaled, const int voxel_wrap[3],
                             uint8_t* color_volume, const uint8_t* colors_rgb24, const float* nmap_curr,
                             int angle_color, int N);
/* returns S = total ray-march steps */
long long kto_raycast(kto_intr intr, const kto_mat33* Rcurr, const float tcurr[3], float tranc_dist,
                      const float volume_size[3], const int16_t* volume, float* vmap, float* nmap,
                      int cols, int rows, const int voxel_wrap[3], uint8_t* vmap_curr_color,
                      const uint8_t* color_volume, int N);
/* axis 0/1/2, back 0/1; elem_size 2 (tsdf) or 4 (colour).  Follows clearVolume{X,Y,Z}{,Back}{,c}. */
void kto_clear_volume(void* vol, int elem_size, int N, int axis, int back, int current_wrap, int delta_wrap);
size_t kto_extract_cloud_slice(const int16_t* volume, const float volume_size[3], kto_point* out, size_t out_cap,
                               const int voxel_wrap[3], const uint8_t* color_volume,
                               int minX, int maxX, int minY, int maxY, int minZ, int maxZ, int subsample,
                               const int real_voxel_wrap[3], int N);

/* ---- f2: the per-slice stage of CloudSliceProcessor (weight cull, VoxelGrid, kNN normals); out48 = 12 floats per point ---- */
size_t kto_slice_process(const kto_point* in, size_t n, int weight_cull, float leaf, int k, float* out48);
/* CloudSliceProcessor::save (backend/CloudSliceProcessor.cpp:180-231): the final VoxelGrid<PointXYZRGBNormal> and the binary PCD */
size_t kto_voxel_grid_normal(const float* in48, size_t n, float leaf, float* out48);
size_t kto_pcd_binary(const float* pts48, size_t n, unsigned char* out);

/* ---- a7 / a10 host math ---- */
void kto_mat33_inverse(const kto_mat33* in, kto_mat33* out);            /* Eigen Matrix3f::inverse() (cofactor) */
void kto_ldlt_solve6(const double A[36], const double b[6], double x[6]); /* Eigen LDLT (pivoted) solve */
void kto_rodrigues(const double r[3], double R[9]);                     /* cv::Rodrigues vector->matrix */
void kto_quat_from_mat33(const kto_mat33* R, float q_xyzw[4]);          /* Eigen Quaternionf(R) */

/* ---- a7, a10, a16: tracker state machine ---- */
typedef struct kto_tracker kto_tracker;
typedef struct {
    int cols, rows, N;
    float fx, fy, cx, cy;
    float volume_size;          /* -s, metres */
    int voxel_shift;            /* -t */
    int overlap;                /* 2, or 0 with -no */
    int static_mode;            /* -sm */
    int use_rgbd, use_rgbd_icp; /* -r, -ri */
    int fast_odometry;          /* -fod */
    int disable_color_angle;    /* -dc */
    int reduce_order;           /* 0 reference float tree, 1 double */
    int dynamic_cube;           /* -d */
    int place_recognition;      /* ConfigArgs::vocabFile.size() != 0 */
} kto_tracker_config;

kto_tracker* kto_tracker_create(const kto_tracker_config* cfg);
void kto_tracker_destroy(kto_tracker* t);
void kto_tracker_reset(kto_tracker* t);
/* KintinuousTracker::repositionCube (KintinuousTracker.cpp:384-442) on explicit state: may move basis[0], basis[2] */
void kto_reposition_cube(const float R[9], const float tlast[3], float volume_size, const float voxel_size[3], int thresh, float basis[3]);
void kto_tracker_get_volume_basis(const kto_tracker* t, float basis[3]);
/* processFrame: depth u16 [rows][cols] mm, rgb24 [rows][cols][3]. */
void kto_tracker_process_frame(kto_tracker* t, const uint16_t* depth, const uint8_t* rgb24, uint64_t timestamp);
/* -p ground-truth odometry (KintinuousTracker::loadTrajectory, GroundTruthOdometry.cpp): pose7 = n x {x y z qx qy qz qw} */
void kto_tracker_load_trajectory(kto_tracker* t, int n, const uint64_t* utimes, const float* pose7);
void kto_tracker_finalise(kto_tracker* t);
/* outputs */
void kto_tracker_get_pose(const kto_tracker* t, float R[9], float tvec[3], float global_cam[3]);
int  kto_tracker_num_poses(const kto_tracker* t);
void kto_tracker_get_dense_pose(const kto_tracker* t, int i, uint64_t* ts, float pose16[16], int* is_loop);
void kto_tracker_get_voxel_wrap(const kto_tracker* t, int wrap[3]);
/* place-recognition tap (KintinuousTracker.cpp:601-624, 706-717, 917-958, 1035-1045): the sampled frames' metadata */
int  kto_tracker_num_pr_samples(const kto_tracker* t);
void kto_tracker_pr_sample(const kto_tracker* t, int i, uint64_t* utime, float trans[3], float rot[9]);
int  kto_tracker_slice_pr_id(const kto_tracker* t, int i);
int  kto_tracker_num_slices(const kto_tracker* t);
size_t kto_tracker_slice_size(const kto_tracker* t, int i);
int  kto_tracker_slice_dimension(const kto_tracker* t, int i);
const kto_point* kto_tracker_slice_points(const kto_tracker* t, int i);
const int16_t* kto_tracker_volume(const kto_tracker* t);
const uint8_t* kto_tracker_color_volume(const kto_tracker* t);
const float* kto_tracker_vmap_g_prev(const kto_tracker* t, int level);
const float* kto_tracker_nmap_g_prev(const kto_tracker* t, int level);
float kto_tracker_trunc_dist(const kto_tracker* t);
/* per-stage accumulated CPU seconds: 0 pyramid, 1 odometry, 2 shift, 3 integrate, 4 raycast, 5 resize */
void kto_tracker_stage_seconds(const kto_tracker* t, double out[6]);
/* statistics of the last frame: U (integrate updates), S (raycast steps) */
void kto_tracker_last_counts(const kto_tracker* t, long long* U, long long* S);

/* test access to the shared atan2 / sin / cos restatement of the per-slice stage (kt_oracle_kernels.c: sp_atan2_pos, sp_sincos) */
float kto_test_sp_atan2_pos(float y, float x);
void kto_test_sp_sincos(float t, float* s, float* c);

#ifdef __cplusplus
}
#endif
#endif
