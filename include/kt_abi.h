/*
 * kt_abi.h -- C-ABI of libkt_hip.so: the MI355X (gfx950) replacement for Kintinuous's device operator
 * API.  The reference has no FFI; its seam is the header src/frontend/cuda/internal.h:295-536 (free
 * functions over DeviceArray2D / PtrStep views, implemented in src/frontend/cuda/ *.cu) plus the
 * container classes in src/frontend/cuda/containers/.  Each entry point below names the reference
 * function it replaces.  kintinuous_amd/host/internal.h re-creates the reference's C++ names as inline
 * wrappers over these calls.
 *
 * Conventions
 *   - every function returns an int status: KT_OK (0) or a KT_ERR_* code; kt_last_error() gives the text.
 *     (The reference prints and exit(0)s on any CUDA error, internal.h:76-86; the C++ wrappers keep that.)
 *   - all image / map / volume pointers are DEVICE pointers (hipMalloc'ed: kt_malloc, or any other HIP
 *     allocation, e.g. a torch tensor's data_ptr) unless the parameter name ends in _host.
 *   - all buffers are dense: row pitch == cols * sizeof(T)  (the reference's volume kernels assume it,
 *     tsdf_volume.cu:612; reduce.cu:444,767).
 *   - vmap / nmap: float[3*rows][cols] (x, y, z planes stacked by rows; invalid = NaN in the x plane).
 *   - tsdf volume: int16[N^3], index x + y*N + z*N*N, storage wrapped by voxel_wrap (tsdf_volume.cu:612);
 *     colour volume: uint8[N^3][4] = r, g, b, weight (the voxel weight lives in .w).
 *   - VOL (internal.h:243) and the image resolution are runtime parameters (N, cols, rows).
 *   - work is enqueued on the context's HIP stream; functions with *_host outputs synchronise that stream
 *     before returning, the others return right after the launch (call kt_sync to wait).
 *   - not thread-safe per context; use one context per host thread / per GPU (the reference is single
 *     GPU-thread too).
 */
#ifndef KT_ABI_H_
#define KT_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KT_OK 0
#define KT_ERR_HIP 1      /* a HIP runtime call or launch failed */
#define KT_ERR_ARG 2      /* bad argument */
#define KT_ERR_NOMEM 3
#define KT_ERR_STATE 4

typedef struct kt_ctx kt_ctx;

typedef struct { float fx, fy, cx, cy; } kt_intr;   /* Intr, internal.h:249-260 */
typedef struct { float m[9]; } kt_mat33;            /* Mat33, internal.h:279-282, row-major */
typedef struct { float v[29]; } kt_jtj;             /* JtJJtrSE3, internal.h:98-149 */
typedef struct {                                    /* DataTerm, internal.h:90-96 (16 bytes) */
    int16_t zero_x, zero_y, one_x, one_y;
    float diff;
    uint8_t valid, pad[3];
} kt_dataterm;
typedef struct {                                    /* PointXYZRGB, internal.h:156-184 (32 bytes) */
    float x, y, z, pad0;
    uint8_t b, g, r, a;
    uint32_t pad1[3];
} kt_point_xyzrgb;
/* internal.h:186-229 == pcl::PointXYZRGBNormal (48 bytes): what CloudSliceProcessor fills CloudSlice::processedCloud with */
typedef struct {
    float x, y, z, pad0;                    /* data[3] = 1 */
    float normal_x, normal_y, normal_z, pad1;
    uint8_t b, g, r, a;
    float curvature;
    float pad2[2];
} kt_point_xyzrgbnormal;

/* ---- context / memory: replaces containers/device_memory.cpp, initialization.cpp, cudaSetDevice ---- */
const char* kt_last_error(void);
const char* kt_version(void);
int kt_device_count(int* count);
int kt_ctx_create(int device, kt_ctx** out);           /* cudaSetDevice(gpu) TrackerInterface.cpp:48 + a private stream */
int kt_ctx_destroy(kt_ctx* ctx);
int kt_ctx_set_stream(kt_ctx* ctx, void* hip_stream);  /* run on a caller-owned hipStream_t (e.g. torch's) */
void* kt_ctx_stream(kt_ctx* ctx);
int kt_sync(kt_ctx* ctx);                              /* cudaDeviceSynchronize at the end of the reference wrappers */
int kt_malloc(kt_ctx* ctx, size_t bytes, void** dptr); /* DeviceMemory::create  device_memory.cpp:98-117 */
int kt_free(kt_ctx* ctx, void* dptr);                  /* DeviceMemory::release */
int kt_memset(kt_ctx* ctx, void* dptr, int value, size_t bytes);
int kt_upload(kt_ctx* ctx, void* dst, const void* src_host, size_t bytes);      /* DeviceMemory::upload */
int kt_download(kt_ctx* ctx, void* dst_host, const void* src, size_t bytes);    /* DeviceMemory::download */
/* DeviceMemory2D::upload/download with a host pitch (device side dense)  device_memory.cpp:206-227 */
int kt_upload2d(kt_ctx* ctx, void* dst, const void* src_host, size_t host_pitch, size_t row_bytes, int rows);
int kt_download2d(kt_ctx* ctx, void* dst_host, size_t host_pitch, const void* src, size_t row_bytes, int rows);

/* ---- image-side kernels ---- */
/* bilateralFilter  internal.h:299 / bilateral_pyrdown.cu:332-342 */
int kt_bilateral_filter(kt_ctx* ctx, const uint16_t* src, uint16_t* dst, int cols, int rows);
/* pyrDown  internal.h:307 / bilateral_pyrdown.cu:344-354; dst is (scols/2) x (srows/2) */
int kt_pyr_down(kt_ctx* ctx, const uint16_t* src, int scols, int srows, uint16_t* dst);
/* createVMap  internal.h:328 / maps.cu:122-137 */
int kt_create_vmap(kt_ctx* ctx, const kt_intr* intr, const uint16_t* depth, int cols, int rows, float* vmap);
/* createNMap  internal.h:335 / maps.cu:139-154 */
int kt_create_nmap(kt_ctx* ctx, const float* vmap, int cols, int rows, float* nmap);
/* tranformMaps  internal.h:346 / maps.cu:203-223 */
int kt_transform_maps(kt_ctx* ctx, const float* vmap_src, const float* nmap_src, int cols, int rows,
                      const kt_mat33* Rmat, const float tvec[3], float* vmap_dst, float* nmap_dst);
/* resizeVMap / resizeNMap  internal.h:453,460 / maps.cu:279-308; out is (in_cols/2) x (in_rows/2) */
int kt_resize_vmap(kt_ctx* ctx, const float* in, int in_cols, int in_rows, float* out);
int kt_resize_nmap(kt_ctx* ctx, const float* in, int in_cols, int in_rows, float* out);
/* generateImage  internal.h:435 / image_generator.cu:56-179: shaded + colour views of a predicted map (vmap_curr_color = the uchar4
 * raycast colour output); dst / dst_color are rgb24 [rows][cols][3].  light_number <= 1 (LightSource holds one position). */
int kt_generate_image(kt_ctx* ctx, const float* vmap, const float* nmap, const uint8_t* vmap_curr_color, int cols, int rows,
                      const float light_pos[3], int light_number, uint8_t* dst_rgb24, uint8_t* dst_color_rgb24);
/* generateDepth  internal.h:442 / image_generator.cu:181-219: depth image (mm) of a predicted map seen from pose (R_inv, t) */
int kt_generate_depth(kt_ctx* ctx, const kt_mat33* R_inv, const float t[3], const float* vmap, const float* nmap, int cols, int rows,
                      uint16_t* dst);
/* shortDepthToMetres  internal.h:313 / bilateral_pyrdown.cu:404-411 */
int kt_depth_to_metres(kt_ctx* ctx, const uint16_t* src, float* dst, int cols, int rows, int cutoff);
/* imageBGRToIntensity  internal.h:315 / bilateral_pyrdown.cu:413-420; src = rgb24 */
int kt_bgr_to_intensity(kt_ctx* ctx, const uint8_t* src_rgb24, uint8_t* dst, int cols, int rows);
/* pyrDownGaussF  internal.h:309 / bilateral_pyrdown.cu:356-377 */
int kt_pyr_down_gauss_f32(kt_ctx* ctx, const float* src, int scols, int srows, float* dst);
/* pyrDownUcharGauss  internal.h:311 / bilateral_pyrdown.cu:379-402 */
int kt_pyr_down_gauss_u8(kt_ctx* ctx, const uint8_t* src, int scols, int srows, uint8_t* dst);
/* computeDerivativeImages  internal.h:301 / bilateral_pyrdown.cu:300-330 */
int kt_derivative_images(kt_ctx* ctx, const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy);
/* projectToPointCloud  internal.h:317-320 / maps.cu:329-344; cloud = float[rows][cols][3]; intrinsics of level 0 */
int kt_project_to_cloud(kt_ctx* ctx, const float* depth, int cols, int rows, float* cloud_xyz,
                        double fx, double fy, double cx, double cy, int level);

/* Fused pyramid build (no single reference counterpart): pyrDown x3 + createVMap x4 + createNMap x4 in one launch,
 * bit-identical to the separate calls (KintinuousTracker.cpp:469-478).  depth0 = bilateral-filtered level 0. */
int kt_build_pyramid(kt_ctx* ctx, const kt_intr* intr, const uint16_t* depth0, int cols, int rows,
                     uint16_t* const depths_out[3], float* const vmaps[4], float* const nmaps[4]);

/* ---- tracking reductions ---- */
/* icpStep  internal.h:485-502 / reduce.cu:347-419.  A_host[36] row-major symmetric, b_host[6], residual_host[2]
 * = {sum residual^2, inlier count}.  The reference's `sum`/`out` scratch arrays live in the context. */
int kt_icp_step(kt_ctx* ctx, const kt_mat33* Rcurr, const float tcurr[3], const float* vmap_curr, const float* nmap_curr,
                const kt_mat33* Rprev_inv, const float tprev[3], const kt_intr* intr,
                const float* vmap_g_prev, const float* nmap_g_prev, int cols, int rows,
                float dist_thres, float angle_thres, float* A_host, float* b_host, float* residual_host);
/* ICPOdometry::getIncrementalTransformation as one call (frontend/ICPOdometry.cpp:68-186; SURVEY 8(b) export list): pose in, pose out.
 * maps[l] = level l of the four pyramids (3 * (rows >> l) planes of (cols >> l) floats; a level with iterations[l] == 0 may be null),
 * intr = level-0 intrinsics (level l uses intr / 2^l, internal.h:255-259), iterations[l] = Gauss-Newton iterations at level l, run
 * from level 3 down to 0 (the reference: {10, 5, 4, 0}, fast odometry {0, 10, 5, 0}, ICPOdometry.cpp:44-55).  Every iteration is a
 * device launch whose epilogue solves the 6x6 system in double (Eigen's pivoted LDLT restated), applies cv::Rodrigues and the
 * SE(3) update (:127-178) and leaves the pose on the device for the next one: no host round trip until the final read-back.
 * A_last_host[36] (may be null) = the last iteration's A, row-major symmetric (the reference keeps it as lastA for getCovariance);
 * residual_host[2] (may be null) = that iteration's {sum residual^2, inlier count}.  Poses are bit-identical to iterating kt_icp_step
 * + kt_host_ldlt_solve6 + kt_host_pose_update on the host. */
int kt_icp_track(kt_ctx* ctx, const float* const vmaps_curr[4], const float* const nmaps_curr[4], const float* const vmaps_g_prev[4],
                 const float* const nmaps_g_prev[4], int cols, int rows, const kt_intr* intr, const kt_mat33* Rprev, const float tprev[3],
                 const int iterations[4], float dist_thres, float angle_thres, kt_mat33* Rcurr_host, float tcurr_host[3],
                 float A_last_host[36], float residual_host[2]);
/* computeRgbResidual  internal.h:519-534 / reduce.cu:798-864 */
int kt_rgb_residual(kt_ctx* ctx, float min_scale, const int16_t* dIdx, const int16_t* dIdy,
                    const float* last_depth, const float* next_depth, const uint8_t* last_image, const uint8_t* next_image,
                    int cols, int rows, kt_dataterm* corres_img, float max_depth_delta, const float kt[3],
                    const kt_mat33* krkinv, int* sigma_sum_host, int* count_host);
/* rgbStep  internal.h:504-517 / reduce.cu:555-607 */
int kt_rgb_step(kt_ctx* ctx, const kt_dataterm* corres_img, float sigma, const float* cloud_xyz, float fx, float fy,
                const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows,
                float* A_host, float* b_host);

/* ---- volume kernels ---- */
/* initVolume / initColorVolume  internal.h:353,417 / tsdf_volume.cu:468-479, 76-87 */
int kt_init_volume(kt_ctx* ctx, int16_t* volume, int N);
int kt_init_color_volume(kt_ctx* ctx, uint8_t* color_volume, int N);
/* integrateTsdfVolume  internal.h:404-409 / tsdf_volume.cu:642-674.  colors = rgb24, nmap_curr = level-0 normal map.
 * Precondition on the volume pair: a state these kernels can leave behind, i.e. grown from kt_init_volume / kt_init_color_volume by
 * integrate and the clears -- weight bytes <= 128 (MAX_WEIGHT).  Any tsdf word and any colour is followed bit for bit
 * (tests/test_gpu_sweep.py::test_random_state_integrate_hip); a foreign volume with a weight byte above 128 keeps that weight where the
 * reference would clamp it to 128 (tests/tools/state_probe.py). */
int kt_integrate_tsdf(kt_ctx* ctx, const uint16_t* depth_raw, int cols, int rows, const kt_intr* intr,
                      const float volume_size[3], const kt_mat33* Rcurr_inv, const float tcurr[3], float tranc_dist,
                      int16_t* volume, float* depth_raw_scaled, const int voxel_wrap[3], uint8_t* color_volume,
                      const uint8_t* colors_rgb24, const float* nmap_curr, int angle_color, int N);
/* raycast  internal.h:429-431 / ray_caster.cu:433-471.  vmap_curr_color = uint8[rows][cols][4] */
int kt_raycast(kt_ctx* ctx, const kt_intr* intr, const kt_mat33* Rcurr, const float tcurr[3], float tranc_dist,
               const float volume_size[3], const int16_t* volume, float* vmap, float* nmap, int cols, int rows,
               const int voxel_wrap[3], uint8_t* vmap_curr_color, const uint8_t* color_volume, int N);
/* clearVolume{X,Y,Z}{,Back}{,c}  internal.h:355-389 / tsdf_volume.cu:117-448.
 * axis 0/1/2 = X/Y/Z, back = the ...Back variants, elem_size 2 = tsdf (short), 4 = colour (uchar4, the ...c variants) */
int kt_clear_volume(kt_ctx* ctx, void* volume, int elem_size, int N, int axis, int back,
                    int current_voxel_wrap, int delta_voxel_wrap);
/* extractCloudSlice  internal.h:463-476 / extract.cu:325-419.  Output order is unspecified (as in the reference). */
int kt_extract_cloud_slice(kt_ctx* ctx, const int16_t* volume, const float volume_size[3], kt_point_xyzrgb* output,
                           size_t output_capacity, const int voxel_wrap[3], const uint8_t* color_volume,
                           int minX, int maxX, int minY, int maxY, int minZ, int maxZ, int subsample,
                           const int real_voxel_wrap[3], int N, size_t* count_host);

/* ---- host math of one Gauss-Newton step (no GPU work), for callers that drive kt_icp_step / kt_rgb_step themselves ----
 * dA.ldlt().solve(db)  ICPOdometry.cpp:130 (Eigen LDLT<6x6 double>: pivoted, pseudo-inverse of D); A row-major */
int kt_host_ldlt_solve6(const double A[36], const double b[6], double x[6]);
/* cv::Rodrigues(rvec, R)  OdometryProvider.h:63 */
int kt_host_rodrigues(const double r[3], double R_out[9]);
/* Eigen Matrix3f::inverse()  ICPOdometry.cpp:77 */
int kt_host_mat33_inverse(const float m[9], float out[9]);
/* resultRt = [Rodrigues(x[3..5]) | x[0..2]] * resultRt;  [Rcurr | tcurr] = [Rprev | tprev] * resultRt^-1 (float)
 * ICPOdometry.cpp:133-178 */
int kt_host_pose_update(const double x[6], double resultRt[16], const float Rprev[9], const float tprev[3],
                        float Rcurr[9], float tcurr[3]);
/* K R K^-1 and K t of resultRt^-1 for computeRgbResidual  RGBDOdometry.cpp:213-231 (rigid Mat::inv(DECOMP_SVD), 3x3 products in double) */
int kt_host_compute_krk(const double resultRt[16], double fx, double fy, double cx, double cy, float krkinv[9], float kt[3]);
/* one line of a -p trajectory file {x y z qx qy qz qw} -> Isometry3f {R row-major (9), t (3)}  KintinuousTracker.cpp:244-256 */
void kt_host_trajectory_pose(const float pose7[7], float T[12]);
/* GroundTruthOdometry::getIncrementalTransformation  GroundTruthOdometry.cpp:42-74: A, B = trajectory poses of the previous and the
 * current frame's stamp, (Rlast, tlast) = rmats_.back(), tvecs_.back() */
void kt_host_ground_truth_pose(const float A[12], const float B[12], const float Rlast[9], const float tlast[3], float Rcurr[9],
                               float tcurr[3]);

/* ---- frame-level tracker (KintinuousTracker::processFrame, KintinuousTracker.cpp:444-915) ----
 * Device-resident fast path: the whole per-frame pipeline is enqueued on the context stream, the ICP /
 * RGB-D Gauss-Newton iterations solve and update the pose on the device (no host round trip per
 * iteration).  The host-side C++ class KintinuousTracker (kintinuous_amd/host) is a thin shell over it. */
typedef struct kt_tracker kt_tracker;
typedef struct {
    int cols, rows, N;
    float fx, fy, cx, cy;
    float volume_size;   /* -s  (ConfigArgs.h:117) */
    int voxel_shift;     /* -t  */
    int overlap;         /* 2 = TrackerInterface::enableOverlap(), 0 with -no */
    int static_mode;     /* -sm */
    int use_rgbd;        /* -r  */
    int use_rgbd_icp;    /* -ri */
    int fast_odometry;   /* -fod */
    int disable_color_angle; /* -dc */
    int max_slice_points;    /* 0 = 3 * cols * rows (KintinuousTracker.cpp:77) */
    int dynamic_cube;        /* -d: the cube swings around the camera with its heading (repositionCube, KintinuousTracker.cpp:384-442);
                                the pose is then observed on the host before the frame is fused (no speculative fusion) */
    int place_recognition;   /* a vocabulary file is configured (-v, ConfigArgs::vocabFile): sample frames for the loop-closure
                                backend whenever the camera has moved (rotation + translation) / 2 >= 0.15 since the last sample, or at
                                the next volume shift (KintinuousTracker.cpp:605-624, 706-717), and mark those poses isLoopPose */
} kt_tracker_config;

int kt_tracker_create(kt_ctx* ctx, const kt_tracker_config* cfg, kt_tracker** out);
int kt_tracker_destroy(kt_tracker* t);
int kt_tracker_reset(kt_tracker* t);
/* processFrame with device-resident inputs: depth u16 [rows][cols] (mm), rgb24 [rows][cols][3] */
int kt_tracker_process_frame(kt_tracker* t, const uint16_t* depth_dev, const uint8_t* rgb24_dev, uint64_t timestamp);
/* Optional read-ahead for log playback: announce a frame that a LATER kt_tracker_process_frame call will receive (same two
 * pointers, contents unchanged until then).  Its pose-independent stages (bilateral filter, depth / vertex / normal pyramids,
 * scaleDepth) run on a second HIP stream, overlapped with the fusion of the frame before it.  Up to two announced frames
 * may be outstanding (a third is refused with KT_ERR_STATE); frames may be handed to kt_tracker_process_frame in any order -- an
 * announced frame is found by its pointers, one that is skipped gives its buffers back.  Best issued right before the
 * kt_tracker_process_frame call of the preceding frame.  Results are identical with and without it. */
int kt_tracker_prefetch_frame(kt_tracker* t, const uint16_t* depth_dev, const uint8_t* rgb24_dev);
/* the same for host-resident frames: pinned staging copy + upload + the pose-independent stages, all on the read-ahead stream.  A
 * later kt_tracker_process_frame_host call with the SAME two host pointers consumes it, so outstanding frames need distinct host
 * buffers (the contents are copied here and may change afterwards). */
int kt_tracker_prefetch_frame_host(kt_tracker* t, const uint16_t* depth_host, const uint8_t* rgb24_host);
/* TrackerInterface::process upload path (TrackerInterface.cpp:90-91): host frame -> device -> processFrame */
int kt_tracker_process_frame_host(kt_tracker* t, const uint16_t* depth_host, const uint8_t* rgb24_host, uint64_t timestamp);
/* -p ground-truth odometry.  KintinuousTracker::loadTrajectory (KintinuousTracker.cpp:216-260) without the text parsing:
 * pose7 = n x {x y z qx qy qz qw}, one row per line "utime,x,y,z,qx,qy,qz,qw" of the trajectory file.  From then on the pose of
 * every frame comes from GroundTruthOdometry (GroundTruthOdometry.cpp:42-74) instead of ICP / RGB-D, and a frame whose timestamp
 * has no entry is dropped (preRun, :89-111).  Call before the first frame; repeated calls add entries. */
int kt_tracker_load_trajectory(kt_tracker* t, int n, const uint64_t* utimes_host, const float* pose7_host);
int kt_tracker_finalise(kt_tracker* t);
/* volumeBasis (KintinuousTracker::getVolumeOffset); constant unless dynamic_cube is set */
int kt_tracker_get_volume_basis(kt_tracker* t, float basis_host[3]);
/* KintinuousTracker::repositionCube on explicit state (host code): may move basis[0] and basis[2] */
/* (|rodrigues2(Rcurr^-1 Rlast)| + |cam - camLast|) / 2, KintinuousTracker.cpp:607-611 */
float kt_host_place_recognition_movement(const float Rcurr[9], const float cam[3], const float Rlast[9], const float camLast[3]);
void kt_host_reposition_cube(const float R[9], const float tlast[3], float volume_size, const float voxel_size[3], int thresh, float basis[3]);
/* rmats_.back() (row-major 3x3), tvecs_.back(), currentGlobalCamera */
int kt_tracker_get_pose(kt_tracker* t, float R_host[9], float t_host[3], float global_cam_host[3]);
int kt_tracker_num_poses(kt_tracker* t);   /* (the kt_tracker_num_* counts are -1 when the frame in flight failed: kt_last_error()) */
/* densePoseGraph[i]: timestamp, row-major 4x4 [R | currentGlobalCamera], isLoopPose */
int kt_tracker_get_dense_pose(kt_tracker* t, int i, uint64_t* ts, float pose16_host[16], int* is_loop);
int kt_tracker_get_voxel_wrap(kt_tracker* t, int wrap_host[3]);
int kt_tracker_num_slices(kt_tracker* t);
int kt_tracker_slice_info(kt_tracker* t, int i, size_t* n_points, int* dimension);
int kt_tracker_slice_points(kt_tracker* t, int i, kt_point_xyzrgb* out_host);
/* the CloudSlice's cameraRotation (row-major), cameraTranslation (= currentGlobalCamera) and utime, CloudSlice.h:110-115 */
int kt_tracker_slice_pose(kt_tracker* t, int i, float R_host[9], float cam_host[3], uint64_t* ts);
/* setParked (KintinuousTracker.cpp:988-991): a parked tracker never shifts */
int kt_tracker_set_parked(kt_tracker* t, int parked);
/* device pointers of the tracker's volume / maps, for inspection and parity tests */
int16_t* kt_tracker_volume(kt_tracker* t);
uint8_t* kt_tracker_color_volume(kt_tracker* t);
float* kt_tracker_vmap_g_prev(kt_tracker* t, int level);
float* kt_tracker_nmap_g_prev(kt_tracker* t, int level);
/* vmap_curr_color: the raycast's uchar4 colour + weight image of the last frame (input of kt_generate_image) */
uint8_t* kt_tracker_vmap_curr_color(kt_tracker* t);
float kt_tracker_trunc_dist(kt_tracker* t);
/* profiling: on = 0 off, 2 = all stages, 1 / 5 / 6 / 4 = only the tsdf23 voxel kernel, on every 8th / 4th / 2nd / every frame (an event
 * pair is two marker packets on the main stream, ~10 us of bubbles: timing every frame lowers the frame rate it is measured next to).
 * kt_tracker_stage_ms returns the MEAN milliseconds per frame since profiling was enabled (hipEvent pairs on
 * the context stream) for: 0 pyramid, 1 odometry, 2 shift, 3 integrate (scaleDepth + tsdf23), 4 raycast,
 * 5 predicted-map resize, 6 the tsdf23 kernel alone; kt_tracker_stage_counts the number of samples of each. */
int kt_tracker_enable_profiling(kt_tracker* t, int on);
int kt_tracker_stage_ms(kt_tracker* t, float ms_host[7]);
int kt_tracker_stage_counts(kt_tracker* t, long long n_host[7]);
/* where the host thread spends a frame: mean seconds per kt_tracker_process_frame call {whole call, waiting for the previous
 * frame's pose}; reset != 0 restarts the statistics */
int kt_tracker_host_times(kt_tracker* t, double out2_host[2], int reset);
/* counters of the last frame (costs two extra syncs per frame; off by default): U = voxels that passed the
 * integrate update predicate, S = ray-march steps */
int kt_tracker_enable_counts(kt_tracker* t, int on);
int kt_tracker_last_counts(kt_tracker* t, unsigned long long* U, unsigned long long* S);
/* Frames whose voxel pass ran from a task plan made ahead of the frame for a predicted pose {hits}, and frames whose pose fell outside
 * the plan's margins and were fused through the in-stream pre-pass instead {misses} (csrc/kt_volume.hip "planning ahead"; results do
 * not depend on which of the two happened). */
int kt_tracker_plan_stats(kt_tracker* t, long long out2_host[2]);
/* Frames whose odometry was run a second time, one launch per iteration, because an inter-workgroup hand-off of the first attempt gave up
 * (csrc/kt_track.hip: kt_icp_level_kernel needs its whole grid resident; another process on the same GPU can keep a workgroup out).  The
 * reference's icpStep is stream-ordered and cannot fail this way (reduce.cu:347-419); here the frame is re-run inside the call that observes
 * it, with the same bits, and only counted.  0 on an undisturbed GPU. */
int kt_tracker_odometry_fallbacks(kt_tracker* t, long long* out_host);


/* The per-slice stage of the backend's CloudSliceProcessor (backend/CloudSliceProcessor.cpp:87-163), the consumer right behind every
 * volume shift: keep points with alpha >= weight_cull (if weight_cull > 0; ConfigArgs::weightCull), pcl::VoxelGrid down-sampling at
 * `leaf` (the voxel size, :124-130), pcl::NormalEstimation with the k (= 20, :148) nearest neighbours, normals flipped towards the sensor
 * origin, output pcl::PointXYZRGBNormal in leaf order.  points_host / out_host are host arrays (a CloudSlice's cloud / processedCloud);
 * out_host needs room for n points; *n_out receives the count. */
int kt_slice_process(kt_ctx* ctx, const kt_point_xyzrgb* points_host, size_t n, int weight_cull, float leaf, int k,
                     kt_point_xyzrgbnormal* out_host, size_t* n_out);
/* The same stage on device-resident points ("slab points are already on device"), asynchronous on the workspace's own
 * stream (or `hip_stream`), with every intermediate -- the input count, the leaf grid, the output count -- kept on the device:
 *   kt_slice_ws_create      buffers for up to `capacity` input points, allocated once;
 *   kt_slice_process_device points_dev[0 .. *n_dev), n_dev a DEVICE word (e.g. the extraction kernel's counter), n_max a host-known
 *                           upper bound of it; enqueues the stage and returns;
 *   kt_slice_ws_count       waits for the stream; *n_out = number of output points, which are kt_slice_ws_output(ws)[0 .. *n_out)
 *                           (device memory, valid until the next call on the workspace).
 * kt_tracker_enable_slice_stage puts it behind the tracker's own shift path. */
typedef struct kt_slice_ws kt_slice_ws;
int kt_slice_ws_create(kt_ctx* ctx, size_t capacity, void* hip_stream, kt_slice_ws** out);
int kt_slice_ws_destroy(kt_slice_ws* ws);
void* kt_slice_ws_stream(kt_slice_ws* ws);
int kt_slice_process_device(kt_slice_ws* ws, const kt_point_xyzrgb* points_dev, const unsigned int* n_dev, size_t n_max, int weight_cull, float leaf, int k);
int kt_slice_ws_count(kt_slice_ws* ws, size_t* n_out);
const kt_point_xyzrgbnormal* kt_slice_ws_output(kt_slice_ws* ws);

/* CloudSliceProcessor::save (backend/CloudSliceProcessor.cpp:180-231), host code, once per run:
 * kt_host_voxel_grid_normal = the pcl::VoxelGrid<pcl::PointXYZRGBNormal> at `leaf` it applies to the concatenated processed clouds when
 * extractOverlap && !saveOverlap (:197-218; every field averaged per leaf, rgb re-packed with a zero alpha byte, leaves in key order);
 * out needs room for n points.  kt_host_save_pcd = pcl::io::savePCDFile(path, cloud, true) (:224-226): PCD v0.7, DATA binary,
 * FIELDS x y z rgb normal_x normal_y normal_z curvature, 32 bytes per point. */
int kt_host_voxel_grid_normal(const kt_point_xyzrgbnormal* in, size_t n, float leaf, kt_point_xyzrgbnormal* out, size_t* n_out);
int kt_host_save_pcd(const char* path, const kt_point_xyzrgbnormal* points, size_t n);

/* The slice stage behind the tracker's own shift path: from this call on every extracted slab also goes through
 * kt_slice_process_device(weight_cull, leaf = the largest voxel edge, k) on a stream of its own (the slab never leaves the device in
 * between), and its CloudSlice::processedCloud travels with the slice.  kt_tracker_slice_processed_info: the number of processed points
 * of slice i, -1 for a slice extracted while the stage was off. */
int kt_tracker_enable_slice_stage(kt_tracker* t, int on, int weight_cull, int k);
int kt_tracker_slice_processed_info(kt_tracker* t, int i, long long* n_points);
int kt_tracker_slice_processed(kt_tracker* t, int i, kt_point_xyzrgbnormal* out);

/* Place-recognition tap (KintinuousTracker::addToPlaceRecognition, KintinuousTracker.cpp:917-958): the frames sampled for the
 * loop-closure backend, in order.  The library keeps the sample's metadata (PlaceRecognitionInput::utime / trans / rotation and the
 * index of the dense pose it belongs to); the caller, who owns the frame buffers, copies the image and depth of that frame
 * (host/KintinuousTracker.h fills placeRecognitionBuffer from them).  slice_pr_id: the sample attached to slice i as its
 * placeRecognitionFrame (mutexOutCloudBuffer's last argument, finalise :1038-1045), or -1. */
int kt_tracker_num_pr_samples(kt_tracker* t);
int kt_tracker_pr_sample(kt_tracker* t, int i, uint64_t* utime, float trans[3], float rotation[9], int* pose_index);
int kt_tracker_slice_pr_id(kt_tracker* t, int i, int* pr_id);

/* ---- multi-GPU: independent streams, one tracker per GPU; poses are gathered by the caller's
 * collective (bench.py / the CLI use RCCL all_gather on the buffer filled here) ---- */
/* copies the last k dense poses (k*16 floats, row-major 4x4) into a DEVICE buffer for the gather */
int kt_tracker_export_poses_device(kt_tracker* t, int k, float* dst_dev);

/* The single collective of the path (north star: "a single RCCL gather of per-stream poses over xGMI"; SURVEY 8(b) export list, 8(e)):
 * one process per GPU, rank r owns stream r.  Rank 0 makes the 128-byte RCCL id and hands it to the other ranks out of band
 * (a file for the C++ driver, torch.distributed's store for bench.py); every rank then joins with kt_comm_init.
 * kt_pose_gather: ONE ncclAllGather of k x 16 floats per rank (row-major [R | currentGlobalCamera], the DensePose payload of
 * KintinuousTracker.h:151-169) on the communicator's own stream; all_poses_host receives nranks * k * 16 floats, rank-major. */
#define KT_COMM_ID_BYTES 128
typedef struct kt_comm kt_comm;
int kt_comm_unique_id(unsigned char id[KT_COMM_ID_BYTES]);
int kt_comm_init(kt_ctx* ctx, int rank, int nranks, const unsigned char id[KT_COMM_ID_BYTES], kt_comm** out);
int kt_pose_gather(kt_comm* comm, kt_tracker* t, int k, float* all_poses_host);
int kt_comm_barrier(kt_comm* comm);   /* returns once every rank has called it (a one-float all-gather) */
int kt_comm_destroy(kt_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* KT_ABI_H_ */
