// kt_internal.hpp -- library-internal (non-ABI) entry points shared between the .hip translation units.
#pragma once

#include "kt_track.hpp"

// Pose of the frame being fused, resident on the device.  The tracker enqueues integrate + raycast before the host knows the pose
// the odometry kernels produced: a small set-up kernel fills this struct (and the z tables) from the device-resident
// Gauss-Newton state; the fusion kernels read it instead of kernel arguments.
struct kt_frame_params {
    float R[9], t[3];   // Rcurr, tcurr (after any host-side shift)
    float Rinv[9];      // Eigen Matrix3f::inverse() of R
    int skip;           // 1: the frame needs a volume shift first -- the speculatively enqueued fusion kernels do nothing
};

// Per-pixel record of integrate, built once per frame by kt_integrate_prepare: everything tsdf23 gathers per voxel from the frame
// (scaled depth with the no-colour sign flag, the colour weight derived from |n_z|, rgb, normal-valid) in ONE 16-byte gather.
struct __attribute__((aligned(16))) kt_pixrec {
    float dp;        // scaleDepth output (negative = "no colour", tsdf_volume.cu:520-527)
    float wrkc;      // (angleColor ? min(1, |n_z| / 0.75) : 1) * 2       tsdf_volume.cu:625
    uint32_t rgbf;   // r | g<<8 | b<<16 | KT_REC_* flags
    uint32_t pad;
};
#define KT_REC_NORMAL_NAN (1u << 24)   // isnan(n_x)
// computeNmapKernel (maps.cu:96-133) writes only n_x = NaN for an invalid normal: n_z keeps whatever the buffer held, and tsdf23
// still reads it for the colour weight of a voxel whose colour is (0, 0, 0).  With ONE nmaps_curr_ buffer, as in the reference,
// that is the n_z of the last frame in which the pixel had a valid normal.  The flag marks such pixels; the tracker, whose
// frame sets rotate, replaces their wrkc from a per-pixel carry kept in processing order (kt_frame_setup_kernel).
#define KT_REC_STALE_NZ (1u << 25)

int kt_bilateral_lut_ensure(kt_ctx* c);
size_t kt_brick_count(int N);   // flags of the raycast's empty-space bricks for an N^3 volume

int kt_integrate_tsdf_impl(kt_ctx* c, const uint16_t* depth_raw, int cols, int rows, const kt_intr* intr,
                           const float volume_size[3], const kt_mat33* Rcurr_inv, const float tcurr[3], float tranc_dist,
                           int16_t* volume, float* depth_raw_scaled, const int voxel_wrap[3], uint8_t* color_volume,
                           const uint8_t* colors, const float* nmap_curr, int angle_color, int N, unsigned int* updated_dev,
                           const void* prepared_rec, const kt_frame_params* fp = nullptr, unsigned char* bricks = nullptr,
                           const float* prepared_dpmax = nullptr);
// device z tables {v_g_z[N], z_scaled[N]} of the next integrate call with a non-null fp (filled by the caller's set-up kernel)
int kt_integrate_tables(kt_ctx* c, int cols, int rows, int N, float** vgz, float** zs);
size_t kt_integrate_rec_bytes(int cols, int rows);
int kt_integrate_prepare(kt_ctx* c, const uint16_t* depth_raw, const uint8_t* colors, const float* nmap_curr, int cols, int rows,
                         const kt_intr* intr, int angle_color, float* depth_raw_scaled, void* rec, float* dpmax);
size_t kt_integrate_dpmax_bytes(void);
int kt_raycast_impl(kt_ctx* c, const kt_intr* intr, const kt_mat33* Rcurr, const float tcurr[3], float tranc_dist,
                    const float volume_size[3], const int16_t* volume, float* vmap, float* nmap, int cols, int rows,
                    const int voxel_wrap[3], uint8_t* vmap_curr_color, const uint8_t* color_volume, int N,
                    unsigned long long* steps_dev, float* const* vpyr, float* const* npyr, const kt_frame_params* fp = nullptr,
                    const unsigned char* bricks = nullptr);
int kt_extract_cloud_slice_async(kt_ctx* c, const int16_t* volume, const float volume_size[3], kt_point_xyzrgb* output,
                                 size_t output_capacity, const int voxel_wrap[3], const uint8_t* color_volume, int minX, int maxX,
                                 int minY, int maxY, int minZ, int maxZ, int subsample, const int real_voxel_wrap[3], int N,
                                 unsigned int* count_dev);
int kt_icp_step_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr,
                       const float* vmap_g_prev, const float* nmap_g_prev, int cols, int rows, float dist_thres, float angle_thres,
                       int mode, const kt_track_state* init = nullptr);
int kt_rgb_residual_device(kt_ctx* c, kt_track_state* state, float min_scale, const int16_t* dIdx, const int16_t* dIdy,
                           const float* last_depth, const float* next_depth, const uint8_t* last_image, const uint8_t* next_image,
                           int cols, int rows, kt_dataterm* corres_img, float max_depth_delta, const uint8_t* cand = nullptr,
                           int write_all = 1);
// the pose-independent half of the residual test, once per frame and level (cand[k] = 1: pixel k can yield a correspondence)
int kt_rgb_residual_candidates(kt_ctx* c, float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* next_depth,
                               const uint8_t* next_image, int cols, int rows, uint8_t* cand);
int kt_joint_step_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr,
                         const float* vmap_g_prev, const float* nmap_g_prev, float dist_thres, float angle_thres,
                         const kt_dataterm* corres_img, const float* cloud, const int16_t* dIdx, const int16_t* dIdy, float sobel_scale,
                         int cols, int rows, const kt_level_k* next_k);
int kt_rgb_step_device(kt_ctx* c, kt_track_state* state, const kt_dataterm* corres_img, const float* cloud, float fx, float fy,
                       const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows, int mode,
                       const kt_level_k* next_k);

// optional HIP events recorded around the tsdf23 voxel kernel (set by the tracker when profiling)
struct kt_event_hook { hipEvent_t ev[2]; bool on; };
extern thread_local kt_event_hook kt_tsdf23_hook;
