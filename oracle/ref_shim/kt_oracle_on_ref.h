/* kt_oracle_on_ref.h -- test infrastructure.  Force-included in front of kt_oracle_host.c (oracle/Makefile, target
 * _ref/libkt_oracle_on_ref.so) it reroutes every kernel call of the oracle's TRACKER (the restated processFrame state machine, odometry
 * loops, shifts) to the reference's own kernels in oracle/_ref/libkt_ref.so.  Running one sequence through libkt_oracle.so and through
 * this hybrid and comparing poses, volumes and slices pins the kernels along the states a real run goes through (wraps after many
 * shifts, volumes hundreds of frames old), not only on constructed inputs (tests/test_oracle_vs_ref.py::test_tracker_on_reference_kernels).
 * The bilateral filter stays the oracle's: its exp() model is the one documented difference. */
#ifndef KT_ORACLE_ON_REF_H
#define KT_ORACLE_ON_REF_H
#include "../kt_oracle.h"

typedef struct { float fx, fy, cx, cy; } ktref_intr;
typedef struct { float m[9]; } ktref_mat33;
void ktref_pyr_down(const uint16_t* src, int scols, int srows, uint16_t* dst);
void ktref_depth_to_metres(const uint16_t* src, float* dst, int cols, int rows, int cutoff);
void ktref_bgr_to_intensity(const uint8_t* src_rgb24, uint8_t* dst, int cols, int rows);
void ktref_pyr_down_gauss_f32(const float* src, int scols, int srows, float* dst);
void ktref_pyr_down_gauss_u8(const uint8_t* src, int scols, int srows, uint8_t* dst);
void ktref_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy);
void ktref_project_to_cloud(const float* depth, int cols, int rows, float* cloud_xyz, double fx, double fy, double cx, double cy, int level);
void ktref_create_vmap(ktref_intr intr, const uint16_t* depth, int cols, int rows, float* vmap);
void ktref_create_nmap(const float* vmap, int cols, int rows, float* nmap);
void ktref_transform_maps(const float* vmap_src, const float* nmap_src, int cols, int rows, const ktref_mat33* R, const float t[3], float* vmap_dst,
                          float* nmap_dst);
void ktref_resize_vmap(const float* in, int in_cols, int in_rows, float* out);
void ktref_resize_nmap(const float* in, int in_cols, int in_rows, float* out);
void ktref_icp_step(const ktref_mat33* Rcurr, const float tcurr[3], const float* vmap_curr, const float* nmap_curr, const ktref_mat33* Rprev_inv,
                    const float tprev[3], ktref_intr intr, const float* vmap_g_prev, const float* nmap_g_prev, int cols, int rows,
                    float dist_thres, float angle_thres, int threads, int blocks, float A[36], float b[6], float residual[2]);
void ktref_rgb_residual(float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* last_depth, const float* next_depth,
                        const uint8_t* last_image, const uint8_t* next_image, int cols, int rows, void* corres, float max_depth_delta,
                        const float kt[3], const ktref_mat33* krkinv, int threads, int blocks, int* sigma_sum, int* count);
void ktref_rgb_step(const void* corres, float sigma, const float* cloud_xyz, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                    float sobel_scale, int cols, int rows, int threads, int blocks, float A[36], float b[6]);
void ktref_init_volume(int16_t* vol, int N);
void ktref_init_color_volume(uint8_t* cvol, int N);
void ktref_integrate_tsdf(const uint16_t* depth_raw, int cols, int rows, ktref_intr intr, const float volume_size[3], const ktref_mat33* Rcurr_inv,
                          const float tcurr[3], float tranc_dist, int16_t* volume, float* depth_scaled, const int voxel_wrap[3],
                          uint8_t* color_volume, const uint8_t* colors_rgb24, const float* nmap_curr, int angle_color, int N);
void ktref_raycast(ktref_intr intr, const ktref_mat33* Rcurr, const float tcurr[3], float tranc_dist, const float volume_size[3],
                   const int16_t* volume, float* vmap, float* nmap, int cols, int rows, const int voxel_wrap[3], uint8_t* vmap_curr_color,
                   const uint8_t* color_volume, int N);
void ktref_clear_volume(void* vol, int elem_size, int N, int axis, int back, int current_wrap, int delta_wrap);
size_t ktref_extract_cloud_slice(const int16_t* volume, const float volume_size[3], void* out, size_t out_cap, const int voxel_wrap[3],
                                 const uint8_t* color_volume, int minX, int maxX, int minY, int maxY, int minZ, int maxZ, int subsample,
                                 const int real_voxel_wrap[3], int N);

static inline ktref_intr refk_intr(kto_intr i) { ktref_intr r = {i.fx, i.fy, i.cx, i.cy}; return r; }
#define REFK_M(p) ((const ktref_mat33*)(p))   /* same layout: nine floats */

static inline void refk_create_vmap(kto_intr intr, const uint16_t* depth, int cols, int rows, float* vmap) { ktref_create_vmap(refk_intr(intr), depth, cols, rows, vmap); }
static inline void refk_transform_maps(const float* vs, const float* ns, int cols, int rows, const kto_mat33* R, const float t[3], float* vd, float* nd)
{ ktref_transform_maps(vs, ns, cols, rows, REFK_M(R), t, vd, nd); }
/* launch geometries: ICPOdometry.cpp / RGBDOdometry.cpp (128 threads; 64 blocks for the steps, 256 for the residual search) */
static inline void refk_icp_step(const kto_mat33* Rc, const float tc[3], const float* vc, const float* nc, const kto_mat33* Rpi, const float tp[3], kto_intr intr,
                                 const float* vp, const float* np_, int cols, int rows, float dt, float at, int order, float A[36], float b[6], float r[2])
{ (void)order; ktref_icp_step(REFK_M(Rc), tc, vc, nc, REFK_M(Rpi), tp, refk_intr(intr), vp, np_, cols, rows, dt, at, 128, 64, A, b, r); }
static inline void refk_rgb_residual(float ms, const int16_t* dx, const int16_t* dy, const float* ld, const float* nd, const uint8_t* li, const uint8_t* ni,
                                     int cols, int rows, kto_dataterm* corres, float mdd, const float kt[3], const kto_mat33* krkinv, int* sigma, int* count)
{ ktref_rgb_residual(ms, dx, dy, ld, nd, li, ni, cols, rows, corres, mdd, kt, REFK_M(krkinv), 128, 256, sigma, count); }
static inline void refk_rgb_step(const kto_dataterm* corres, float sigma, const float* cloud, float fx, float fy, const int16_t* dx, const int16_t* dy,
                                 float sobel, int cols, int rows, int order, float A[36], float b[6])
{ (void)order; ktref_rgb_step(corres, sigma, cloud, fx, fy, dx, dy, sobel, cols, rows, 128, 64, A, b); }
static inline long long refk_integrate_tsdf(const uint16_t* d, int cols, int rows, kto_intr intr, const float vs[3], const kto_mat33* Ri, const float t[3],
                                            float trunc, int16_t* vol, float* scaled, const int wrap[3], uint8_t* col, const uint8_t* rgb,
                                            const float* nmap, int angle, int N)
{ ktref_integrate_tsdf(d, cols, rows, refk_intr(intr), vs, REFK_M(Ri), t, trunc, vol, scaled, wrap, col, rgb, nmap, angle, N); return 0; }   /* (U is a diagnostic) */
static inline long long refk_raycast(kto_intr intr, const kto_mat33* R, const float t[3], float trunc, const float vs[3], const int16_t* vol, float* vmap,
                                     float* nmap, int cols, int rows, const int wrap[3], uint8_t* cm, const uint8_t* col, int N)
{ ktref_raycast(refk_intr(intr), REFK_M(R), t, trunc, vs, vol, vmap, nmap, cols, rows, wrap, cm, col, N); return 0; }
static inline size_t refk_extract_cloud_slice(const int16_t* vol, const float vs[3], kto_point* out, size_t cap, const int wrap[3], const uint8_t* col,
                                              int x0, int x1, int y0, int y1, int z0, int z1, int sub, const int real[3], int N)
{ return ktref_extract_cloud_slice(vol, vs, out, cap, wrap, col, x0, x1, y0, y1, z0, z1, sub, real, N); }

#define kto_pyr_down ktref_pyr_down
#define kto_depth_to_metres ktref_depth_to_metres
#define kto_bgr_to_intensity ktref_bgr_to_intensity
#define kto_pyr_down_gauss_f32 ktref_pyr_down_gauss_f32
#define kto_pyr_down_gauss_u8 ktref_pyr_down_gauss_u8
#define kto_derivative_images ktref_derivative_images
#define kto_project_to_cloud ktref_project_to_cloud
#define kto_create_vmap refk_create_vmap
#define kto_create_nmap ktref_create_nmap
#define kto_transform_maps refk_transform_maps
#define kto_resize_vmap ktref_resize_vmap
#define kto_resize_nmap ktref_resize_nmap
#define kto_icp_step refk_icp_step
#define kto_rgb_residual refk_rgb_residual
#define kto_rgb_step refk_rgb_step
#define kto_init_volume ktref_init_volume
#define kto_init_color_volume ktref_init_color_volume
#define kto_integrate_tsdf refk_integrate_tsdf
#define kto_raycast refk_raycast
#define kto_clear_volume ktref_clear_volume
#define kto_extract_cloud_slice refk_extract_cloud_slice
#endif
