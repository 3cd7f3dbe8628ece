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
#ifndef KT_REC_BYTES
#define KT_REC_BYTES 12   // 16 (a padding word, one aligned 16-byte gather): +27 % fetched bytes on the 768^3 case, +8 % on the orbit, +2.5 % / 0 % time (profiles/r03_tsdf23_pmc_variants_call4.log)
#endif
#if KT_REC_BYTES == 16
struct __attribute__((aligned(16))) kt_pixrec {
#else
struct kt_pixrec {
#endif
    float dp;        // scaleDepth output (negative = "no colour", tsdf_volume.cu:520-527)
    float wrkc;      // (angleColor ? min(1, |n_z| / 0.75) : 1) * 2       tsdf_volume.cu:625
    uint32_t rgbf;   // r | g<<8 | b<<16 | KT_REC_* flags
#if KT_REC_BYTES == 16
    uint32_t pad;
#endif
};
static_assert(sizeof(kt_pixrec) == KT_REC_BYTES, "pixel record size");
#define KT_REC_NORMAL_NAN (1u << 24)   // isnan(n_x)
// computeNmapKernel (maps.cu:96-133) writes only n_x = NaN for an invalid normal: n_z keeps whatever the buffer held, and tsdf23
// still reads it for the colour weight of a voxel whose colour is (0, 0, 0).  With ONE nmaps_curr_ buffer, as in the reference,
// that is the n_z of the last frame in which the pixel had a valid normal.  The flag marks such pixels; the tracker, whose
// frame sets rotate, replaces their wrkc from a per-pixel carry kept in processing order (kt_frame_setup_kernel).
#define KT_REC_STALE_NZ (1u << 25)

// Checkpoint of tsdf23's incremental walk for one storage column (sx, sy): the reference advances v_x, v_y by repeated float `+=`
// from z = 0 (tsdf_volume.cu:566-574, quirk A.17: the values are DEFINED by that recurrence), so a task that starts at z = zc needs
// the walked values there.  zc is wave-uniform at both call sites (kt_tsdf_interval_kernel, kt_frame_setup_kernel).
__device__ __forceinline__ float2 kt_tsdf_walk_checkpoint(const float* Ri, float tx, float ty, float tz, float cell_x, float cell_y, float cell_z,
                                                           float fx, float fy, int sx, int sy, int wx, int wy, int N, int zc)
{
    int x = sx - wx; if (x < 0) x += N;
    int y = sy - wy; if (y < 0) y += N;
    const float v_g_x = __builtin_fmaf((float)x + 0.5f, cell_x, -tx);
    const float v_g_y = __builtin_fmaf((float)y + 0.5f, cell_y, -ty);
    const float v_g_z0 = __builtin_fmaf(0 + 0.5f, cell_z, -tz);
    float v_x = __builtin_fmaf(Ri[2], v_g_z0, __builtin_fmaf(Ri[0], v_g_x, Ri[1] * v_g_y)) * fx;
    float v_y = __builtin_fmaf(Ri[5], v_g_z0, __builtin_fmaf(Ri[3], v_g_x, Ri[4] * v_g_y)) * fy;
    const float dvx = Ri[2] * cell_z * fx, dvy = Ri[5] * cell_z * fy;
    int z = 0;
    for (; z + 16 <= zc; z += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { v_x += dvx; v_y += dvy; }
    }
    for (; z < zc; ++z) { v_x += dvx; v_y += dvy; }
    return make_float2(v_x, v_y);
}

int kt_bilateral_lut_ensure(kt_ctx* c);
size_t kt_brick_count(int N);   // flags of the raycast's empty-space bricks for an N^3 volume

int kt_integrate_tsdf_impl(kt_ctx* c, const uint16_t* depth_raw, int cols, int rows, const kt_intr* intr,
                           const float volume_size[3], const kt_mat33* Rcurr_inv, const float tcurr[3], float tranc_dist,
                           int16_t* volume, float* depth_raw_scaled, const int voxel_wrap[3], uint8_t* color_volume,
                           const uint8_t* colors, const float* nmap_curr, int angle_color, int N, unsigned int* updated_dev,
                           const void* prepared_rec, const kt_frame_params* fp = nullptr, unsigned char* bricks = nullptr,
                           const float* prepared_dpmax = nullptr, const struct kt_tsdf_plan* plan = nullptr);
// One frame's work list for the voxel kernel (kt_volume.hip "planning ahead"): wave-column z-ranges, the compact task list, and the walk
// checkpoints of the active wave-columns.  kt_integrate_plan fills ranges and tasks for a PREDICTED pose with margins theta (rad) and
// tau (m) on a stream of the caller's choice; the checkpoints need the frame's own pose (kt_tsdf_walk_checkpoint, set-up kernel).
struct kt_tsdf_plan { unsigned int* wrange; unsigned int* tasks; unsigned int* task_count; float2* walk0; };
int kt_tsdf_plan_alloc(kt_tsdf_plan* p, int N);
void kt_tsdf_plan_free(kt_tsdf_plan* p);
void kt_tsdf_plan_shape(int cols, int rows, int N, int* wx, int* wy, int* xg, int* yg);   // wave-column shape and grid of these launches (for the checkpoint workgroups)
#define KT_NO_PLAN (-1)   // kt_integrate_plan: no plan can be made for these margins (not an error: the caller takes the in-stream pre-pass)
int kt_integrate_plan(hipStream_t stream, const kt_tsdf_plan* plan, const void* rec, const float* dpmax, int cols, int rows, const kt_intr* intr,
                      const float volume_size[3], const kt_mat33* Rinv_pred, const float t_pred[3], float tranc_dist, const int voxel_wrap[3], int N,
                      float theta, float tau);
// device z tables {v_g_z[N], z_scaled[N]} of the next integrate call with a non-null fp (filled by the caller's set-up kernel)
int kt_integrate_tables(kt_ctx* c, int cols, int rows, int N, float** vgz, float** zs);
size_t kt_integrate_rec_bytes(int cols, int rows);
int kt_integrate_prepare(kt_ctx* c, const uint16_t* depth_raw, const uint8_t* colors, const float* nmap_curr, int cols, int rows,
                         const kt_intr* intr, int angle_color, float* depth_raw_scaled, void* rec, float* dpmax);
size_t kt_integrate_dpmax_bytes(int cols, int rows);
struct kt_pyr_args;
int kt_pyr_args_fill(kt_pyr_args* a, const kt_intr* intr, const uint16_t* depth0, int cols, int rows, uint16_t* const depths_out[3], float* const vmaps[4],
                     float* const nmaps[4]);
int kt_pyramid01_launch(kt_ctx* c, const kt_pyr_args* a);
int kt_frame_prepare(kt_ctx* c, const kt_intr* intr, const uint16_t* depth_filtered, uint16_t* const depths_out[3], float* const vmaps[4], float* const nmaps[4],
                     const uint16_t* depth_raw, const uint8_t* colors, int cols, int rows, int angle_color, float* depth_raw_scaled, void* rec, float* dpmax);
int kt_raycast_impl(kt_ctx* c, const kt_intr* intr, const kt_mat33* Rcurr, const float tcurr[3], float tranc_dist,
                    const float volume_size[3], const int16_t* volume, float* vmap, float* nmap, int cols, int rows,
                    const int voxel_wrap[3], uint8_t* vmap_curr_color, const uint8_t* color_volume, int N,
                    unsigned long long* steps_dev, float* const* vpyr, float* const* npyr, const kt_frame_params* fp = nullptr,
                    const unsigned char* bricks = nullptr);
int kt_extract_cloud_slice_async(kt_ctx* c, const int16_t* volume, const float volume_size[3], kt_point_xyzrgb* output,
                                 size_t output_capacity, const int voxel_wrap[3], const uint8_t* color_volume, int minX, int maxX,
                                 int minY, int maxY, int minZ, int maxZ, int subsample, const int real_voxel_wrap[3], int N,
                                 unsigned int* count_dev);
int kt_refill_granules(kt_ctx* c);   // kt_track.hip: every hand-off granule back to the sentinel, on the context's stream
int kt_icp_step_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr,
                       const float* vmap_g_prev, const float* nmap_g_prev, int cols, int rows, float dist_thres, float angle_thres,
                       int mode, const kt_track_state* init = nullptr, int keep29 = 0);
int kt_icp_level_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr, const float* vmap_g_prev,
                        const float* nmap_g_prev, int cols, int rows, float dist_thres, float angle_thres, const kt_track_state* frame, int first, int n_iter);
int kt_icp_levels_device(kt_ctx* c, kt_track_state* state, int n_levels, const float* const* vmaps_curr, const float* const* nmaps_curr, const kt_intr* intrs,
                         const float* const* vmaps_g_prev, const float* const* nmaps_g_prev, const int* cols, const int* rows, const int* n_iter,
                         float dist_thres, float angle_thres, const kt_track_state* frame, int first,
                         const struct kt_setup_args* fused_setup = nullptr);   // the levels of a frame in ONE launch (+ its set-up in the epilogue: kt_setup.hpp)
bool kt_ri_levels_selected();             // -ri: one launch per pyramid level (kt_joint_level_kernel); off by default
bool kt_icp_levels_forced();               // ... asked for explicitly
bool kt_icp_levels_selected(int device);   // kt_track.hip: KT_ICP_LEVELS / kt_debug_icp_levels, and the device can hold the whole grid
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
int kt_joint_level_device(kt_ctx* c, kt_track_state* state, const float* vmap_curr, const float* nmap_curr, const kt_intr* intr, const float* vmap_g_prev,
                          const float* nmap_g_prev, float dist_thres, float angle_thres, kt_dataterm* corres_img, const float* cloud, const int16_t* dIdx,
                          const int16_t* dIdy, float sobel_scale, float min_scale, const float* last_depth, const float* next_depth, const uint8_t* last_image,
                          const uint8_t* next_image, float max_depth_delta, const uint8_t* cand, int cols, int rows, int n_iter, const kt_level_k* k_level,
                          const kt_level_k* k_next);   // the -ri iterations of one pyramid level in ONE launch (kt_track.hip: kt_joint_level_kernel)
int kt_rgb_step_device(kt_ctx* c, kt_track_state* state, const kt_dataterm* corres_img, const float* cloud, float fx, float fy,
                       const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows, int mode,
                       const kt_level_k* next_k);

// kt_slice.hip: a device array of `*n_dev` items (clamped to cap) and its count into (pinned) destination memory, on `st`
int kt_copy_counted(hipStream_t st, const void* src, void* dst, const unsigned int* n_dev, unsigned int cap, int item_bytes, unsigned int* count_out);
const unsigned int* kt_slice_ws_leaves_dev(kt_slice_ws* w);   // device word: output count of the workspace's last call

// optional HIP events recorded around the tsdf23 voxel kernel (set by the tracker when profiling)
struct kt_event_hook { hipEvent_t ev[2]; bool on; };
extern thread_local kt_event_hook kt_tsdf23_hook;
