// kt_tracker.hip -- KintinuousTracker::processFrame (src/frontend/KintinuousTracker.cpp:444-915) as a
// device-resident pipeline on one HIP stream: pyramid build -> odometry (all Gauss-Newton iterations
// enqueued back to back, solved on the device) -> ONE host sync for the pose -> cyclical-volume shift
// (slab extract + clear) -> integrate -> raycast -> predicted-map pyramid.  The reference performs 19
// blocking host round trips per frame (reduce.cu:398-401); this path performs one.
// Host-side state (pose lists, shift decision, slices, dense pose graph) restates
// KintinuousTracker.cpp:71-182, 262-382, 575-667, 1003-1048, 1075-1085, 1156-1208.
#include "kt_internal.hpp"
#include "kt_setup.hpp"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <deque>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <atomic>
#include <map>
#include <thread>
#include <vector>

namespace {

struct DensePose { uint64_t ts; float pose[16]; int is_loop; };   // KintinuousTracker.h:151-169
struct Slice {      // CloudSlice.h:28-129 (cloud + dimension; processed = CloudSlice::processedCloud when the slice stage is on)
    std::vector<kt_point_xyzrgb> pts; int dim; float R[9], cam[3]; uint64_t ts; int pr_id;
    std::vector<kt_point_xyzrgbnormal> processed; bool has_processed = false;
};
struct PrSample { uint64_t utime; float trans[3], rot[9]; int pose_index; };   // PlaceRecognitionInput.h:30-56, minus the frame bytes

// Everything computed from the input frame alone (bilateral + pyramids + scaleDepth records): two sets, so the set of frame
// k + 1 can be filled on the prefetch stream while frame k is tracked and fused on the main stream.
struct FrameSet {
    uint16_t* depths[KT_LEVELS];
    float *vmaps[KT_LEVELS], *nmaps[KT_LEVELS];
    void* rec;            // per-pixel integrate records (kt_integrate_prepare)
    float* dpmax;         // per 32 x 32 pixel tile: largest |scaled depth| (interval pre-pass prune)
    float* scaled;        // depthRawScaled_
    // RGB-D odometry inputs derived from the frame alone (RGBDOdometry.cpp:140-158, 186, 296-300); a frame is "next" while it is
    // tracked and "last" for its successor, so the sets double as RGBDOdometry's lastDepth / nextDepth ... buffers
    float* depth_m[KT_LEVELS];        // metric depth pyramid
    uint8_t* image[KT_LEVELS];        // intensity pyramid
    int16_t *dIdx[KT_LEVELS], *dIdy[KT_LEVELS];
    uint8_t* cand[KT_LEVELS];         // pixels that can yield a photometric correspondence when the frame is "next" (pose-independent tests)
    float* cloud[KT_LEVELS];          // projectToPointCloud of depth_m (used when the frame is "last")
    hipEvent_t ready;     // recorded on the prefetch stream when the set is complete
    long long user;       // ordinal of the process_frame call that last consumed the set (-1: none)
};
// Three sets rotate: the frame whose fusion is still running (also RGB-D "last" of its successor), the frame about to be tracked
// and the frame being read ahead.  A set is recycled behind odo_ev (wait_frame_consumed).
#define KT_NSETS 3

typedef kt_pose_mirror PoseMirror;   // csrc/kt_setup.hpp: the host's window on the frame in flight
struct Pending { const uint16_t* depth; const uint8_t* rgb; int set; const uint16_t* depth_host; const uint8_t* rgb_host; };
#define KT_NSLOTS 4   // host-frame staging: frame in flight + two read-aheads + the one being filled
#define KT_NODO 8     // ring of "odometry of frame f enqueued" events

#ifndef KT_SIDE_GATE_DEFAULT
#define KT_SIDE_GATE_DEFAULT 2
#endif
enum { ST_PYRAMID = 0, ST_ODOMETRY, ST_SHIFT, ST_INTEGRATE, ST_RAYCAST, ST_RESIZE, ST_TSDF23, ST_COUNT };

}  // namespace

static std::atomic<int> kt_live_trackers{0};   // live trackers of this process: the level form of the ICP chain needs to be alone (kt_tracker_create)
struct kt_tracker {
    kt_ctx* ctx;
    kt_tracker_config cfg;
    kt_intr intr;
    int N;
    float volume_size[3], voxel_size[3];
    float tranc_dist;
    float volume_basis[3];
    float initial_rotation[9];
    int voxel_wrap[3], v_wrap_copy[3];
    int global_time;
    uint64_t current_ts = 0;
    // -p ground-truth odometry: camera_trajectory (KintinuousTracker.h:237; its comparator is std::less<int>, so the keys are the
    // timestamps narrowed to int) and current_utime as GroundTruthOdometry sees it (the previous tracked frame's timestamp)
    bool has_trajectory = false;
    std::map<int, std::array<float, 12>> trajectory;   // R row-major [0..8], t [9..11]
    uint64_t gt_utime = 0;
    bool parked;
    float Rlast[9], tlast[3];          // rmats_.back(), tvecs_.back()
    float current_global_camera[3];
    // device buffers (KintinuousTracker.h:185-222)
    int16_t* tsdf; uint8_t* color;
    uint16_t* depths_curr[KT_LEVELS];
    float *vmaps_curr[KT_LEVELS], *nmaps_curr[KT_LEVELS], *vmaps_g_prev[KT_LEVELS], *nmaps_g_prev[KT_LEVELS];
    uint8_t* vmap_curr_color;
    float* depth_raw_scaled;
    void* rec_curr;                    // integrate records of the current frame (set member, like the *_curr maps above)
    // the *_curr pointers above alias sets[cur_set]
    FrameSet sets[KT_NSETS];
    int last_assigned;                 // set handed to the most recent frame (prefetched or inline)
    std::vector<Pending> pending;      // prefetched frames not yet processed (at most 2)
    long long frames_started;          // process_frame calls so far
    long long frames_observed;         // 1 + ordinal of the latest frame whose pose the host has seen in the mirror (complete_frame)
    long long out_ordinal;             // ordinal of the frame in flight (t->outstanding)
    hipEvent_t guard_ev;               // only for out-of-pattern read-aheads (see kt_tracker_prefetch_frame)
    // odo_ev[f % KT_NODO]: recorded on the main stream between fusion(f - 1) and fusion(f), in -p mode only -- there the host never
    // waits for a pose, so a lagging GPU's fusion could otherwise read a frame set the read-ahead stream has recycled (wait_frame_consumed).
    hipEvent_t odo_ev[8];
    long long slot_frame[KT_NSLOTS];           // ordinal of the frame that consumed staging slot k (-1: none)
    hipStream_t pre_stream;
    kt_ctx pre_ctx;                    // the context with pre_stream as its stream (image kernels only)
    size_t cloud_cap;
    // RGBDOdometry buffers (RGBDOdometry.h:96-111)
    int prev_set;                      // set of the previous frame ("last" of RGBDOdometry), -1 before the first frame
    kt_dataterm* corres[KT_LEVELS];
    // device-resident Gauss-Newton state + pinned mirror
    kt_track_state* state_dev; kt_track_state* state_host;
    // staging for the host-frame entry point
    // KT_NSLOTS rotating slots: pinned host copy + device copy of one frame, an event recorded after its upload
    uint16_t* depth_stage[KT_NSLOTS]; uint8_t* rgb_stage[KT_NSLOTS];
    uint16_t* depth_stage_host[KT_NSLOTS]; uint8_t* rgb_stage_host[KT_NSLOTS];  // pinned
    hipEvent_t slot_uploaded[KT_NSLOTS];
    int next_slot;
    // outputs
    std::vector<DensePose> poses;
    std::deque<Slice> slices;          // a deque: the download thread fills slices[i].pts while new slices may be appended
    // Slice download off the frame's critical path.  The extraction kernel writes the points into pinned host memory (two buffers,
    // alternating), the count follows by an asynchronous copy and an event; a helper thread waits for the event and copies the n points
    // into the slice while the main thread has long gone on enqueueing the clears and the fusion.  Getters join (join_slice_jobs).
    // ONE worker thread per tracker, started at creation (where it also pays HIP's per-thread set-up): a thread per shift put its
    // creation and, worse, its first hipSetDevice -- which takes the runtime's lock for up to a millisecond -- into the shift frame
    // (frame period of the first shift of a run: 1.44 ms against 0.47-0.56 later; r03 call 16).
    struct SliceJob { bool active; bool done; size_t slice; hipError_t status; };
    SliceJob jobs[2];
    std::thread worker;
    std::mutex wmu;
    std::condition_variable wcv, wdone;
    std::deque<std::function<void()>> wq;
    bool wstop;
    kt_point_xyzrgb* cloud_host[2]; unsigned int* cloud_count_host[2]; hipEvent_t cloud_ev[2];
    int cloud_next;
    // The CloudSliceProcessor stage behind a shift, on the device (kt_slice.hip; kt_tracker_enable_slice_stage): the extraction then
    // writes into device memory, and a stream of its own takes the slab from there -- raw points to pinned host memory, the stage, the
    // processed points to pinned host memory -- while the main stream goes on with the clears and the fusion.
    bool slice_stage; int slice_cull, slice_k;
    kt_slice_ws* slice_ws;
    kt_point_xyzrgb* cloud_dev[2]; unsigned int* cloud_count_dev[2]; hipEvent_t extracted[2];
    kt_point_xyzrgbnormal* proc_host[2]; unsigned int* proc_count_host[2];
    // place-recognition tap (KintinuousTracker.h:216, 248-249): pose of the last sampled frame, the samples
    float pr_rot[9], pr_trans[3];
    std::vector<PrSample> pr_samples;
    // deferred completion: a frame's fusion kernels are enqueued before the host has seen its pose (complete_frame)
    bool outstanding; uint64_t out_ts; int out_set;
    const uint16_t* out_depth; const uint8_t* out_rgb; int out_thresh;
    bool out_speculated;               // the fusion kernels of the frame in flight were enqueued against the device's shift decision
    kt_frame_params* fp_dev;
    // colour weight carried across frames for pixels without a valid normal (KT_REC_STALE_NZ): [carry_sel] = state before the frame
    // in flight, [carry_sel ^ 1] = state after it
    float* wrkc_carry[2]; int carry_sel;
    // Planning ahead (kt_volume.hip): the voxel kernel's task plan of a read-ahead frame is made for a PREDICTED pose on plan_stream at the start of
    // the frame's kt_tracker_process_frame call -- before its odometry is enqueued, one increment past the last pose the host has seen -- and has the
    // odometry launch to finish.  plan_sel: the slot the frame in flight was enqueued with, -1 = none (the in-stream pre-pass ran).
    // (set / depth / rgb / wrap: what the plan was built from -- kept for drop_plans_of_set and the debug state)
    struct PlanSlot { kt_tsdf_plan plan; hipEvent_t done; long long ordinal; float R[9], t[3], theta, tau; int wrap[3]; int set; const uint16_t* depth; const uint8_t* rgb; };
    bool icp_levels;   // ICP-only odometry: one launch per pyramid level (kt_icp_level_kernel) instead of one per iteration, while this tracker is alone
    bool last_icp_levels;   // ... and whether the last frame's chain took that form
    bool counted;           // this tracker is in kt_live_trackers (a create that failed early is not)
    // A hand-off time-out is not the end of the frame (round 6): complete_frame re-runs the frame's odometry in the stepwise form, and the
    // level form stays off for icp_demote more frames (doubling per relapse: something -- another process on the GPU, a CU-masked queue --
    // keeps its grid from being resident; the stepwise chain needs no co-residency and gives the same bits).
    int icp_demote, icp_demote_len;
    long long odo_fallbacks;   // frames whose odometry was re-run (kt_tracker_odometry_fallbacks)
    int out_last_set;          // RGB-D "last" set of the frame in flight (for that re-run)
    bool setup_fused;          // the frame's set-up ran in the epilogue of its odometry launch (kt_icp_level_kernel): no kt_frame_setup_kernel was enqueued
    // Side-stream gate (round 6).  The read-ahead of frame f + 1 is enqueued the moment the host has seen the pose of frame f - 1 -- exactly
    // when that frame's voxel kernel starts -- and the voxel kernel is a fixed grid of 8192 waves that fills EVERY wave slot of the chip and
    // deals its task list statically over them: one foreign wave on one SIMD keeps one of its workgroups out until another has finished, and
    // the launch is as long as its slowest SIMD (farwall768: 0.49 ms by the kernel trace with nothing beside it, 0.67 with a single sleeping
    // wave of another stream resident -- profiles/r06_experiments.md, call 4; 0.85 next to the 1280x960 read-ahead).  With the gate on, the
    // read-ahead stream waits for an event recorded between the voxel kernel and the ray cast (one marker packet on the main stream: 3-5 us of
    // a millisecond frame), so the read-ahead runs beside the ray cast -- whose workgroups are handed out dynamically -- and the next odometry.
    // (A first cut had the side stream start with a one-thread kernel sleeping on a device flag the ray cast set -- no packet on the main
    // stream; that resident wave was itself the disturbance.)
    hipEvent_t gate_ev; bool gate_armed; int side_gate;   // side_gate: 0 off, 1 on
    PlanSlot plans[3]; hipStream_t plan_stream;          // (three: frame f is planned while the voxel kernels of f - 1 and f - 2 may still read theirs)
    int plan_sel; bool plan_enabled; float plan_margin_scale;
    // test hooks (kt_tracker_debug_plan_*): the pose every frame WILL arrive at, taken from an identical earlier run, as the prediction;
    // an offset of fr * theta about a random axis and ft * tau along a random direction on top of the prediction; fixed margins; a log
    // of the poses the set-up kernels saw
    std::vector<float> plan_truth; float plan_fr = 0.0f, plan_ft = 0.0f, plan_theta_fixed = 0.0f, plan_tau_fixed = 0.0f; bool plan_perturb = false;
    unsigned int plan_rng = 0x9e3779b9u; std::vector<float> pose_log; bool pose_log_on = false;
    float hist_R[2][9], hist_gc[2][3]; int hist_n;      // rotation and global camera of the two most recent tracked frames
    float pred_err_t, pred_err_r;                       // error of the most recent prediction (metres, radians): drives the margins
    long long plan_hits, plan_misses;
    unsigned char* bricks;             // negative-brick flags of the volume (tsdf23 raises, raycast skips; kt_volume.hip)
    float *vgz_dev, *zs_dev;           // z tables of integrate (kt_integrate_tables)
    PoseMirror* mirror;                // pinned + mapped host memory
    unsigned int frame_seq;            // sequence number of the frame in flight (PoseMirror::seq)
    long long prof_frames;             // frames seen with profiling == 1 / 5 (the tsdf23 event pair is recorded on every 8th / 4th)
    // profiling / counters (events double-buffered by frame parity: a pair is read one frame after it was recorded)
    double host_wait_s, host_call_s; long long host_calls;   // where the host thread spends a frame (kt_tracker_host_times)
    int profiling; int ev_par;
    hipEvent_t ev[2][ST_COUNT][2]; bool ev_rec[2][ST_COUNT];
    double stage_ms_sum[ST_COUNT]; long long stage_n[ST_COUNT]; float stage_ms_last[ST_COUNT];
    int counting;
    unsigned int* upd_dev; unsigned long long* steps_dev;
    unsigned long long last_U, last_S;
};

static int join_slice_jobs(kt_tracker* t);
static int plan_frame(kt_tracker* t, const struct Pending& frame, long long ordinal);
static int lvl_cols(const kt_tracker* t, int l) { return t->cfg.cols >> l; }
static int lvl_rows(const kt_tracker* t, int l) { return t->cfg.rows >> l; }
static kt_intr lvl_intr(kt_intr k, int l)  // Intr::operator() internal.h:255-259
{
    const int div = 1 << l;
    kt_intr r = {k.fx / div, k.fy / div, k.cx / div, k.cy / div};
    return r;
}


// zero-fill goes on the context's stream: the stream is non-blocking, so a null-stream hipMemset would not be ordered
// against the kernels that later write these buffers
static thread_local hipStream_t g_alloc_stream = nullptr;
template <typename T>
static int dev_alloc(T** p, size_t count, bool zero)
{
    KT_HIP(hipMalloc((void**)p, (count ? count : 1) * sizeof(T)));
    if (zero) KT_HIP(hipMemsetAsync(*p, 0, (count ? count : 1) * sizeof(T), g_alloc_stream));
    return KT_OK;
}

// make sets[q] the current frame's set
static void select_set(kt_tracker* t, int q)
{
    for (int l = 0; l < KT_LEVELS; ++l) {
        t->depths_curr[l] = t->sets[q].depths[l];
        t->vmaps_curr[l] = t->sets[q].vmaps[l];
        t->nmaps_curr[l] = t->sets[q].nmaps[l];
    }
    t->depth_raw_scaled = t->sets[q].scaled;
    t->rec_curr = t->sets[q].rec;
}

// next set in rotation that is neither the previous frame's (its fusion may still run; RGB-D "last" of the coming frame) nor owned
// by an outstanding read-ahead; -1 if there is none
static int pick_free_set(kt_tracker* t)
{
    for (int k = 1; k <= KT_NSETS; ++k) {
        const int q = (t->last_assigned + k) % KT_NSETS;
        bool taken = q == t->prev_set;
        for (const Pending& p : t->pending) taken = taken || p.set == q;
        if (!taken) return q;
    }
    return -1;
}

// Events that only order one stream of this device behind another: no timing, no system-scope fence (the cache write-back of the
// default record is a 6 us bubble on the main stream; kernel boundaries already order device memory between streams of one agent).
#ifndef KT_EV_DEVICE
#define KT_EV_DEVICE (hipEventDisableTiming | hipEventDisableSystemFence)
#endif

// `stream` may overwrite what frame `u` consumed (its frame set, its staging slot) only after fusion(u) and the RGB-D "last" reads of
// odometry(u + 1).  Three ways to know, cheapest first:
//   1. the host has seen the pose of frame u + 1 in the mirror: the set-up kernel that posted it runs behind odometry(u + 1), which
//      runs behind fusion(u) -- the normal case (kt_tracker_prefetch_frame observes the frame in flight first), no stream operation;
//   2. -p mode, where the host never waits for a pose: odo_ev[u + 1], recorded between fusion(u) and fusion(u + 1);
//   3. otherwise order against everything enqueued so far.
// (A record per frame in every mode is a 5 us bubble on the main stream, rocprofv3 kernel trace: a marker packet between two kernels.)
static int wait_frame_consumed(kt_tracker* t, hipStream_t stream, long long u)
{
    if (u < 0 || stream == t->ctx->stream) return KT_OK;   // the main stream is ordered by itself
    if (u + 2 <= t->frames_observed) return KT_OK;
    if (t->has_trajectory && u + 1 < t->frames_started) {
        // a slot of the ring that has since been re-recorded by frame u + 1 + k * KT_NODO only waits longer, never less
        KT_HIP(hipStreamWaitEvent(stream, t->odo_ev[(u + 1) % KT_NODO], 0));
    } else {
        KT_HIP(hipEventRecord(t->guard_ev, t->ctx->stream));
        KT_HIP(hipStreamWaitEvent(stream, t->guard_ev, 0));
    }
    return KT_OK;
}

// minimumGradientMagnitudes / sobelScale of RGBDOdometry (RGBDOdometry.cpp:64-70, 236): the squared-gradient threshold of level l
static float rgbd_min_scale(int l)
{
    const float min_grad[KT_LEVELS] = {12, 5, 3, 1};
    const double SOBEL_SCALE = 1.0 / pow(2.0, 3);
    return (float)(pow(min_grad[l], 2.0) / pow(SOBEL_SCALE, 2.0));
}

// [A] of processFrame (KintinuousTracker.cpp:465-479) plus the pose-independent half of integrate, into sets[q] on cx's stream
static int build_frame_set(kt_tracker* t, kt_ctx* cx, int q, const uint16_t* depth_raw, const uint8_t* colors)
{
    const int cols = t->cfg.cols, rows = t->cfg.rows;
    FrameSet& fs = t->sets[q];
    // the odometry provider objects of the ctor (KintinuousTracker.cpp:128-178): ground truth wins over the RGB-D flags
    const bool icp = !t->has_trajectory && !(t->cfg.use_rgbd || t->cfg.use_rgbd_icp);
    const bool rgbd = !t->has_trajectory && !icp;
    static const bool fused_prepare = []() { const char* e = getenv("KT_PREPARE_FUSED"); return e ? atoi(e) != 0 : true; }();
    if (icp || t->cfg.use_rgbd_icp || !t->cfg.disable_color_angle) {
        KT_TRY(kt_bilateral_filter(cx, depth_raw, fs.depths[0], cols, rows));
        uint16_t* dl[3] = {fs.depths[1], fs.depths[2], fs.depths[3]};
        if (fused_prepare)   // pyramid + scaleDepth + pixel records in three launches (kt_volume.hip)
            KT_TRY(kt_frame_prepare(cx, &t->intr, fs.depths[0], dl, fs.vmaps, fs.nmaps, depth_raw, colors, cols, rows, !t->cfg.disable_color_angle, fs.scaled, fs.rec, fs.dpmax));
        else {
            KT_TRY(kt_build_pyramid(cx, &t->intr, fs.depths[0], cols, rows, dl, fs.vmaps, fs.nmaps));
            KT_TRY(kt_integrate_prepare(cx, depth_raw, colors, fs.nmaps[0], cols, rows, &t->intr, !t->cfg.disable_color_angle, fs.scaled, fs.rec, fs.dpmax));
        }
    } else
        KT_TRY(kt_integrate_prepare(cx, depth_raw, colors, fs.nmaps[0], cols, rows, &t->intr, !t->cfg.disable_color_angle, fs.scaled, fs.rec, fs.dpmax));
    if (rgbd) {
        // RGBDOdometry::populateRGBDData (RGBDOdometry.cpp:140-158), the derivative images of the frame as "next" (:296-300) and its
        // point clouds as "last" (:186): all functions of the frame alone
        KT_TRY(kt_depth_to_metres(cx, depth_raw, fs.depth_m[0], cols, rows, (int)(6.0 * 1000)));
        for (int l = 0; l + 1 < KT_LEVELS; ++l) KT_TRY(kt_pyr_down_gauss_f32(cx, fs.depth_m[l], lvl_cols(t, l), lvl_rows(t, l), fs.depth_m[l + 1]));
        KT_TRY(kt_bgr_to_intensity(cx, colors, fs.image[0], cols, rows));
        for (int l = 0; l + 1 < KT_LEVELS; ++l) KT_TRY(kt_pyr_down_gauss_u8(cx, fs.image[l], lvl_cols(t, l), lvl_rows(t, l), fs.image[l + 1]));
        const double ifx = t->intr.fx, ify = t->intr.fy, icx = t->intr.cx, icy = t->intr.cy;  // RGBDOdometry.cpp:72-75
        for (int l = 0; l < KT_LEVELS; ++l) {
            KT_TRY(kt_derivative_images(cx, fs.image[l], lvl_cols(t, l), lvl_rows(t, l), fs.dIdx[l], fs.dIdy[l]));
            KT_TRY(kt_rgb_residual_candidates(cx, rgbd_min_scale(l), fs.dIdx[l], fs.dIdy[l], fs.depth_m[l], fs.image[l], lvl_cols(t, l),
                                              lvl_rows(t, l), fs.cand[l]));
            KT_TRY(kt_project_to_cloud(cx, fs.depth_m[l], lvl_cols(t, l), lvl_rows(t, l), fs.cloud[l], ifx, ify, icx, icy, l));
        }
    }
    return KT_OK;
}

static void compute_global_camera(kt_tracker* t, const float* tcurr)
{
    // KintinuousTracker.cpp:581-595 (and :274-287 in reset())
    for (int k = 0; k < 3; ++k) {
        const float initial_trans = (float)((double)t->volume_basis[k] - (double)t->cfg.volume_size * 0.5);
        t->current_global_camera[k] = initial_trans;
        t->current_global_camera[k] += (float)t->voxel_wrap[k] * t->voxel_size[k];
        if (tcurr) t->current_global_camera[k] += tcurr[k] - t->volume_basis[k];
    }
}

static void v_wrap_copy_update(kt_tracker* t)
{
    // vWrapCopyUpdate KintinuousTracker.cpp:1075-1085
    for (int k = 0; k < 3; ++k) {
        t->v_wrap_copy[k] = t->voxel_wrap[k];
        if (t->v_wrap_copy[k] < 0) t->v_wrap_copy[k] = t->N - ((-t->v_wrap_copy[k]) % t->N);
    }
}

static void push_pose(kt_tracker* t, uint64_t ts, const float* R, int is_loop)
{
    DensePose p;
    p.ts = ts;
    p.is_loop = is_loop;
    for (int i = 0; i < 16; ++i) p.pose[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) p.pose[i * 4 + j] = R[i * 3 + j];
        p.pose[i * 4 + 3] = t->current_global_camera[i];
    }
    t->poses.push_back(p);
}

// ---- profiling helpers ---------------------------------------------------------------------------
// profiling modes: 0 off; 1 the tsdf23 event pair on every 8th frame (long timed regions: the timing stays off the other 7);
// 2 every stage of every frame (serial breakdown pass); 4 the tsdf23 pair on EVERY frame (short timed regions)
static bool prof_all(const kt_tracker* t) { return t->profiling == 2 || t->profiling == 3; }
static bool prof_tsdf(const kt_tracker* t) { return t->profiling == 1 || t->profiling == 4 || t->profiling == 5 || t->profiling == 6; }
static int ev_begin(kt_tracker* t, int st)
{
    if (prof_all(t) || (prof_tsdf(t) && st == ST_TSDF23)) {
        KT_HIP(hipEventRecord(t->ev[t->ev_par][st][0], t->ctx->stream));
    }
    return KT_OK;
}
static int ev_end(kt_tracker* t, int st)
{
    if (prof_all(t) || (prof_tsdf(t) && st == ST_TSDF23)) {
        KT_HIP(hipEventRecord(t->ev[t->ev_par][st][1], t->ctx->stream));
        t->ev_rec[t->ev_par][st] = true;
    }
    return KT_OK;
}
static void tsdf23_hook_arm(kt_tracker* t)
{
    // 4: every frame; 1 / 5: every 8th / 4th (an event pair costs two marker packets = ~10 us of bubbles on the main stream)
    kt_tsdf23_hook.on = prof_all(t) || t->profiling == 4 || (t->profiling == 1 && (t->prof_frames++ % 8) == 0) || (t->profiling == 5 && (t->prof_frames++ % 4) == 0) ||
                        (t->profiling == 6 && (t->prof_frames++ % 2) == 0);   // 6: every second frame (the driver's 20-frame region: 10 samples)
    kt_tsdf23_hook.ev[0] = t->ev[t->ev_par][ST_TSDF23][0];
    kt_tsdf23_hook.ev[1] = t->ev[t->ev_par][ST_TSDF23][1];
    if (kt_tsdf23_hook.on) t->ev_rec[t->ev_par][ST_TSDF23] = true;
}
// harvest every recorded pair whose end event has completed (pairs still in flight stay armed)
static void ev_collect(kt_tracker* t)
{
    if (!t->profiling) return;
    for (int par = 0; par < 2; ++par)
        for (int st = 0; st < ST_COUNT; ++st)
            if (t->ev_rec[par][st] && hipEventQuery(t->ev[par][st][1]) == hipSuccess) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, t->ev[par][st][0], t->ev[par][st][1]) == hipSuccess) {
                    t->stage_ms_sum[st] += ms;
                    t->stage_n[st] += 1;
                    t->stage_ms_last[st] = ms;
                }
                t->ev_rec[par][st] = false;
            }
}

extern "C" {

static int tracker_create_impl(kt_tracker* t, kt_ctx* ctx, const kt_tracker_config* cfg);

int kt_tracker_create(kt_ctx* ctx, const kt_tracker_config* cfg, kt_tracker** out)
{
    KT_ARG(ctx && cfg && out);
    KT_ARG(cfg->cols > 0 && cfg->rows > 0 && cfg->N > 0 && cfg->volume_size > 0);
    KT_ARG((cfg->cols % 8) == 0 && (cfg->rows % 8) == 0);  // 4 pyramid levels
    KT_ARG(cfg->N <= 1536);                                 // 32-bit voxel offsets (kt_volume.hip)
    KT_HIP(hipSetDevice(ctx->device));
    kt_tracker* t = new kt_tracker();   // value-initialised: every pointer starts null, so a failed create can be destroyed
    const int s = tracker_create_impl(t, ctx, cfg);
    if (s != KT_OK) {
        t->ctx = ctx;
        (void)kt_tracker_destroy(t);   // kt_last_error() keeps the message of the failure
        return s;
    }
    *out = t;
    return KT_OK;
}

static int tracker_create_impl(kt_tracker* t, kt_ctx* ctx, const kt_tracker_config* cfg)
{
    t->ctx = ctx;
    g_alloc_stream = ctx->stream;
    t->cfg = *cfg;
    t->N = cfg->N;
    // KintinuousTracker ctor, KintinuousTracker.cpp:71-182
    t->intr.fx = cfg->fx; t->intr.fy = cfg->fy; t->intr.cx = cfg->cx; t->intr.cy = cfg->cy;
    for (int k = 0; k < 3; ++k) {
        t->volume_size[k] = cfg->volume_size;
        t->voxel_size[k] = cfg->volume_size / (float)cfg->N;
    }
    for (int k = 0; k < 9; ++k) t->initial_rotation[k] = (k % 4 == 0) ? 1.f : 0.f;
    for (int k = 0; k < 3; ++k) t->volume_basis[k] = t->volume_size[k] * 0.5f;
    if (cfg->static_mode || cfg->dynamic_cube)  // :101-110
        t->volume_basis[2] = t->volume_size[2] * 0.5f - (float)(((double)t->volume_size[2] * 0.5) + (cfg->static_mode ? 0.45 : 0));
    const float default_tranc = fmaxf(0.01f, t->volume_size[0] / 100.0f);               // :112
    const float mc = fmaxf(t->voxel_size[0], fmaxf(t->voxel_size[1], t->voxel_size[2]));
    t->tranc_dist = fmaxf(default_tranc, 2.1f * mc);                                    // TSDFVolume.cpp:89-97
    t->voxel_wrap[0] = t->voxel_wrap[1] = t->voxel_wrap[2] = 0;

    const size_t nvox = (size_t)cfg->N * cfg->N * cfg->N;
    const size_t P = (size_t)cfg->cols * cfg->rows;
    KT_TRY(dev_alloc(&t->tsdf, nvox, false));
    KT_TRY(dev_alloc(&t->color, nvox * 4, false));
    for (int l = 0; l < KT_LEVELS; ++l) {  // allocateBuffers :356-382; zero-filled so "stale" planes are defined
        const size_t p = (size_t)lvl_cols(t, l) * lvl_rows(t, l);
        for (int q = 0; q < KT_NSETS; ++q) {
            KT_TRY(dev_alloc(&t->sets[q].depths[l], p, true));
            KT_TRY(dev_alloc(&t->sets[q].vmaps[l], 3 * p, true));
            KT_TRY(dev_alloc(&t->sets[q].nmaps[l], 3 * p, true));
        }
        KT_TRY(dev_alloc(&t->vmaps_g_prev[l], 3 * p, true));
        KT_TRY(dev_alloc(&t->nmaps_g_prev[l], 3 * p, true));
        const bool rgbd = cfg->use_rgbd || cfg->use_rgbd_icp;
        const size_t q = rgbd ? p : 0;
        for (int sidx = 0; sidx < KT_NSETS; ++sidx) {
            KT_TRY(dev_alloc(&t->sets[sidx].depth_m[l], q, true));
            KT_TRY(dev_alloc(&t->sets[sidx].image[l], q, true));
            KT_TRY(dev_alloc(&t->sets[sidx].dIdx[l], q, true));
            KT_TRY(dev_alloc(&t->sets[sidx].dIdy[l], q, true));
            KT_TRY(dev_alloc(&t->sets[sidx].cand[l], q, true));
            KT_TRY(dev_alloc(&t->sets[sidx].cloud[l], 3 * q, true));
        }
        KT_TRY(dev_alloc(&t->corres[l], q, true));
    }
    KT_TRY(dev_alloc(&t->vmap_curr_color, P * 4, true));
    for (int q = 0; q < KT_NSETS; ++q) {
        KT_TRY(dev_alloc(&t->sets[q].scaled, P, true));
        unsigned char* rec = nullptr;
        KT_TRY(dev_alloc(&rec, kt_integrate_rec_bytes(cfg->cols, cfg->rows), true));
        t->sets[q].rec = rec;
        unsigned char* dpm = nullptr;
        KT_TRY(dev_alloc(&dpm, kt_integrate_dpmax_bytes(cfg->cols, cfg->rows), true));
        t->sets[q].dpmax = (float*)dpm;
        KT_HIP(hipEventCreateWithFlags(&t->sets[q].ready, KT_EV_DEVICE));
        t->sets[q].user = -1;
    }
    t->frames_started = 0;
    t->cloud_next = 0;
    t->slice_stage = false; t->slice_ws = nullptr;
    for (int b = 0; b < 2; ++b) { t->cloud_dev[b] = nullptr; t->cloud_count_dev[b] = nullptr; t->extracted[b] = nullptr; t->proc_host[b] = nullptr; t->proc_count_host[b] = nullptr; }
    for (int b = 0; b < 2; ++b) { t->jobs[b].active = false; t->jobs[b].done = true; t->jobs[b].status = hipSuccess; t->cloud_host[b] = nullptr; t->cloud_count_host[b] = nullptr; t->cloud_ev[b] = nullptr; }
    t->wstop = false;
    {
        const int device = ctx->device;
        t->worker = std::thread([t, device]() {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lk(t->wmu);
            for (;;) {
                t->wcv.wait(lk, [t]() { return t->wstop || !t->wq.empty(); });
                if (t->wq.empty()) return;   // stop requested and nothing left to do
                std::function<void()> job = std::move(t->wq.front());
                t->wq.pop_front();
                lk.unlock();
                job();
                lk.lock();
                t->wdone.notify_all();
            }
        });
    }
    t->frames_observed = 0;
    t->out_ordinal = -1;
    KT_HIP(hipEventCreateWithFlags(&t->guard_ev, KT_EV_DEVICE));
    for (int k = 0; k < KT_NODO; ++k) KT_HIP(hipEventCreateWithFlags(&t->odo_ev[k], KT_EV_DEVICE));
    for (int k = 0; k < KT_NSLOTS; ++k) t->slot_frame[k] = -1;
    KT_HIP(hipStreamCreateWithFlags(&t->pre_stream, hipStreamNonBlocking));
    KT_TRY(kt_bilateral_lut_ensure(ctx));   // before the context is cloned: both streams share the table
    t->pre_ctx = *ctx;
    t->pre_ctx.stream = t->pre_stream;
    t->pre_ctx.own_stream = false;
    t->last_assigned = KT_NSETS - 1;
    select_set(t, 0);
    t->cloud_cap = cfg->max_slice_points > 0 ? (size_t)cfg->max_slice_points : P * 3;  // cloud_device_(numPixels * 3) :77
    for (int b = 0; b < 2; ++b) {
        KT_HIP(hipHostMalloc((void**)&t->cloud_host[b], t->cloud_cap * sizeof(kt_point_xyzrgb), hipHostMallocDefault));
        KT_HIP(hipHostMalloc((void**)&t->cloud_count_host[b], sizeof(unsigned int), hipHostMallocDefault));
        KT_HIP(hipEventCreateWithFlags(&t->cloud_ev[b], hipEventDisableTiming));
    }
    KT_TRY(dev_alloc(&t->state_dev, 1, true));
    KT_HIP(hipHostMalloc((void**)&t->state_host, sizeof(kt_track_state), hipHostMallocDefault));
    for (int k = 0; k < KT_NSLOTS; ++k) {
        KT_TRY(dev_alloc(&t->depth_stage[k], P, true));
        KT_TRY(dev_alloc(&t->rgb_stage[k], P * 3, true));
        KT_HIP(hipHostMalloc((void**)&t->depth_stage_host[k], P * sizeof(uint16_t), hipHostMallocDefault));
        KT_HIP(hipHostMalloc((void**)&t->rgb_stage_host[k], P * 3, hipHostMallocDefault));
        KT_HIP(hipEventCreateWithFlags(&t->slot_uploaded[k], hipEventDisableTiming));
    }
    t->next_slot = 0;
    KT_TRY(dev_alloc(&t->upd_dev, 16, true));
    KT_TRY(dev_alloc(&t->steps_dev, 4, true));
    t->profiling = 0;
    t->counting = 0;
    for (int par = 0; par < 2; ++par)
        for (int s = 0; s < ST_COUNT; ++s) {
            KT_HIP(hipEventCreate(&t->ev[par][s][0]));
            KT_HIP(hipEventCreate(&t->ev[par][s][1]));
            t->ev_rec[par][s] = false;
        }
    t->ev_par = 0;
    t->outstanding = false;
    t->host_wait_s = t->host_call_s = 0.0; t->host_calls = 0;
    KT_TRY(dev_alloc(&t->fp_dev, 1, true));
    KT_HIP(hipStreamCreateWithFlags(&t->plan_stream, hipStreamNonBlocking));
    // The level form is used only while this is the ONLY live tracker of the process (decided per frame, icp_odometry): its workgroups wait for
    // each other inside a launch and need the whole machine, so anything that keeps compute units busy next to it for long -- a second tracker
    // fed from the same host (scripts/multistream_one_gpu.py) -- can keep its last workgroup out until the bounded waits give up.  The
    // tracker's own side streams are finite per frame and its main-stream kernels are ordered behind the launch.  What this process cannot
    // see -- another process on the same GPU -- is handled per frame: a launch whose waits give up (50 ms) aborts, complete_frame re-runs the
    // frame's odometry in the stepwise form and keeps the level form off for a while (icp_demote).
    t->icp_levels = kt_icp_levels_selected(ctx->device);
    t->last_icp_levels = false;
    t->icp_demote = 0; t->icp_demote_len = 64; t->odo_fallbacks = 0; t->out_last_set = -1;
    kt_live_trackers.fetch_add(1);
    t->counted = true;
    for (int k = 0; k < 3; ++k) {
        KT_TRY(kt_tsdf_plan_alloc(&t->plans[k].plan, cfg->N));
        KT_HIP(hipEventCreateWithFlags(&t->plans[k].done, KT_EV_DEVICE));
        t->plans[k].ordinal = -1; t->plans[k].set = -1; t->plans[k].depth = nullptr; t->plans[k].rgb = nullptr;
    }
    KT_HIP(hipEventCreateWithFlags(&t->gate_ev, KT_EV_DEVICE));
    t->gate_armed = false;
    {
        // KT_SIDE_GATE: 0 never, 1 always, 2 (default) on dense views only -- the rule that picks 32 x 2 wave-columns (more than 1.5 pixels per
        // voxel column: the voxel kernel is the frame's longest and its launch is dense)
        const char* e = getenv("KT_SIDE_GATE");
        const int mode = e ? atoi(e) : KT_SIDE_GATE_DEFAULT;
        int wcx = 0, wcy = 0, xg = 0, yg = 0;
        kt_tsdf_plan_shape(cfg->cols, cfg->rows, cfg->N, &wcx, &wcy, &xg, &yg);
        // "dense view": the voxel pass's rule for 32 x 2 wave-columns AND an image beyond 640x480 -- the case where the voxel kernel is the frame's
        // longest and the read-ahead is heavy (1280x960 into 768^3).  640x480 into 256^3 is dense by the first rule alone, but its voxel kernel is
        // 21 us and its frame is the orbit's: gated and stepwise it ran at 3770 frames/s against 4100 (profiles/r06_experiments.md, call 12)
        const bool dense = wcx == 32 && (long long)cfg->cols * cfg->rows > 640LL * 480LL;
        t->side_gate = mode == 1 || (mode == 2 && dense) ? 1 : 0;
        // A level launch takes every compute unit's whole register file: whatever the side streams have not finished when the odometry starts
        // waits for it to end and then runs beside the voxel kernel after all.  On a dense view (1280x960: the pixel loops are long, the
        // hand-over a level launch saves is 1.4 % of the frame) the odometry therefore stays one launch per iteration -- 48 VGPRs: read-ahead
        // and plan run UNDER it -- and the plan's completion event, which the set-up kernel already waits for, keeps both out of the voxel kernel.
        const char* d = getenv("KT_DENSE_STEPWISE");
        if (mode == 2 && dense && (d ? atoi(d) != 0 : true) && !kt_icp_levels_forced()) t->icp_levels = false;
    }
    t->plan_enabled = getenv("KT_NO_PLAN") == nullptr;   // (A/B switch: every frame through the in-stream pre-pass)
    t->plan_margin_scale = getenv("KT_PLAN_MARGIN_SCALE") ? (float)atof(getenv("KT_PLAN_MARGIN_SCALE")) : 1.0f;   // (tests: 0 makes every plan miss)
    t->plan_hits = t->plan_misses = 0;
    KT_TRY(dev_alloc(&t->bricks, kt_brick_count(cfg->N) + 16, true));
    for (int k = 0; k < 2; ++k) KT_TRY(dev_alloc(&t->wrkc_carry[k], (size_t)cfg->cols * cfg->rows, true));  // zero: the oracle's calloc'ed normal map
    t->carry_sel = 0;
    KT_TRY(kt_integrate_tables(ctx, cfg->cols, cfg->rows, cfg->N, &t->vgz_dev, &t->zs_dev));
    KT_HIP(hipHostMalloc((void**)&t->mirror, sizeof(PoseMirror), hipHostMallocMapped | hipHostMallocCoherent));
    memset(t->mirror, 0, sizeof(PoseMirror));
    t->frame_seq = 0;
    t->prof_frames = 0;
    KT_TRY(kt_tracker_reset(t));
    // The kernels of the shift path run for the first time HERE, on the empty volume (a one-voxel extraction, one plane of zeros
    // cleared to zero), not in the middle of the first shift frame: a kernel's first launch pays the runtime's lazy set-up for it.
    {
        const int zero[3] = {0, 0, 0};
        KT_TRY(kt_extract_cloud_slice_async(ctx, t->tsdf, t->volume_size, t->cloud_host[0], t->cloud_cap, zero, t->color, 0, 1, 0, 1, 0, 1, 1, zero, t->N,
                                            &ctx->counters[1]));
        KT_TRY(kt_clear_volume(ctx, t->tsdf, 2, t->N, 0, 0, 0, 0));
        KT_TRY(kt_clear_volume(ctx, t->color, 4, t->N, 0, 0, 0, 0));
        KT_HIP(hipStreamSynchronize(ctx->stream));
    }
    return KT_OK;
}

int kt_tracker_destroy(kt_tracker* t)
{
    if (!t) return KT_OK;
    if (t->counted) kt_live_trackers.fetch_sub(1);   // (advisor, round 5: a create that failed before it was counted took the counter to -1)
    (void)hipStreamSynchronize(t->ctx->stream);
    (void)join_slice_jobs(t);
    if (t->worker.joinable()) {
        { std::lock_guard<std::mutex> lk(t->wmu); t->wstop = true; }
        t->wcv.notify_all();
        t->worker.join();
    }
    (void)hipFree(t->tsdf); (void)hipFree(t->color);
    for (int l = 0; l < KT_LEVELS; ++l) {
        for (int q = 0; q < KT_NSETS; ++q) { (void)hipFree(t->sets[q].depths[l]); (void)hipFree(t->sets[q].vmaps[l]); (void)hipFree(t->sets[q].nmaps[l]); }
        (void)hipFree(t->vmaps_g_prev[l]); (void)hipFree(t->nmaps_g_prev[l]);
        for (int sidx = 0; sidx < KT_NSETS; ++sidx) {
            (void)hipFree(t->sets[sidx].depth_m[l]); (void)hipFree(t->sets[sidx].image[l]); (void)hipFree(t->sets[sidx].dIdx[l]);
            (void)hipFree(t->sets[sidx].dIdy[l]); (void)hipFree(t->sets[sidx].cloud[l]); (void)hipFree(t->sets[sidx].cand[l]);
        }
        (void)hipFree(t->corres[l]);
    }
    if (t->pre_stream) (void)hipStreamSynchronize(t->pre_stream);
    for (int q = 0; q < KT_NSETS; ++q) {
        (void)hipFree(t->sets[q].scaled); (void)hipFree(t->sets[q].rec); (void)hipFree(t->sets[q].dpmax);
        if (t->sets[q].ready) (void)hipEventDestroy(t->sets[q].ready);
    }
    if (t->guard_ev) (void)hipEventDestroy(t->guard_ev);
    for (int k = 0; k < KT_NODO; ++k)
        if (t->odo_ev[k]) (void)hipEventDestroy(t->odo_ev[k]);
    if (t->pre_stream) (void)hipStreamDestroy(t->pre_stream);
    (void)hipFree(t->vmap_curr_color);
    for (int b = 0; b < 2; ++b) {
        (void)hipHostFree(t->cloud_host[b]); (void)hipHostFree(t->cloud_count_host[b]);
        if (t->cloud_ev[b]) (void)hipEventDestroy(t->cloud_ev[b]);
        (void)hipFree(t->cloud_dev[b]); (void)hipFree(t->cloud_count_dev[b]); (void)hipHostFree(t->proc_host[b]); (void)hipHostFree(t->proc_count_host[b]);
        if (t->extracted[b]) (void)hipEventDestroy(t->extracted[b]);
    }
    if (t->slice_ws) (void)kt_slice_ws_destroy(t->slice_ws);
    (void)hipFree(t->state_dev); (void)hipHostFree(t->state_host);
    for (int k = 0; k < KT_NSLOTS; ++k) {
        (void)hipFree(t->depth_stage[k]); (void)hipFree(t->rgb_stage[k]);
        (void)hipHostFree(t->depth_stage_host[k]); (void)hipHostFree(t->rgb_stage_host[k]);
        if (t->slot_uploaded[k]) (void)hipEventDestroy(t->slot_uploaded[k]);
    }
    (void)hipFree(t->upd_dev); (void)hipFree(t->steps_dev);
    for (int par = 0; par < 2; ++par)
        for (int s = 0; s < ST_COUNT; ++s) {
            if (t->ev[par][s][0]) (void)hipEventDestroy(t->ev[par][s][0]);
            if (t->ev[par][s][1]) (void)hipEventDestroy(t->ev[par][s][1]);
        }
    (void)hipHostFree(t->mirror);
    if (t->plan_stream) { (void)hipStreamSynchronize(t->plan_stream); (void)hipStreamDestroy(t->plan_stream); }
    for (int k = 0; k < 3; ++k) {
        kt_tsdf_plan_free(&t->plans[k].plan);
        if (t->plans[k].done) (void)hipEventDestroy(t->plans[k].done);
    }
    (void)hipFree(t->fp_dev);
    if (t->gate_ev) (void)hipEventDestroy(t->gate_ev);
    (void)hipFree(t->bricks);
    for (int k = 0; k < 2; ++k) (void)hipFree(t->wrkc_carry[k]);
    delete t;
    return KT_OK;
}

int kt_tracker_reset(kt_tracker* t)
{
    // KintinuousTracker::reset :262-354
    KT_ARG(t);
    t->outstanding = false;
    KT_TRY(kt_refill_granules(t->ctx));   // whatever a failed or abandoned frame left in the hand-off buffer
    KT_HIP(hipStreamSynchronize(t->ctx->stream));
    t->global_time = 0;
    memcpy(t->Rlast, t->initial_rotation, sizeof(t->Rlast));
    memcpy(t->tlast, t->volume_basis, sizeof(t->tlast));
    compute_global_camera(t, nullptr);
    t->voxel_wrap[0] = t->voxel_wrap[1] = t->voxel_wrap[2] = 0;
    t->poses.clear();
    (void)join_slice_jobs(t);
    t->slices.clear();
    t->pr_samples.clear();
    memcpy(t->pr_trans, t->current_global_camera, sizeof(t->pr_trans));   // :290-291
    memcpy(t->pr_rot, t->initial_rotation, sizeof(t->pr_rot));
    KT_HIP(hipStreamSynchronize(t->pre_stream));
    t->pending.clear();
    KT_HIP(hipStreamSynchronize(t->plan_stream));
    t->plan_sel = -1;
    for (int k = 0; k < 3; ++k) { t->plans[k].ordinal = -1; t->plans[k].set = -1; }
    t->hist_n = 0;
    t->pred_err_t = t->pred_err_r = 0.0f;
    t->prev_set = -1;
    t->parked = t->cfg.static_mode != 0;
    t->v_wrap_copy[0] = t->v_wrap_copy[1] = t->v_wrap_copy[2] = 0;
    t->gt_utime = 0;
    t->current_ts = 0;
    // the colour-weight carry of pixels without a normal starts from zero, like the freshly allocated nmaps_curr_ it stands for
    for (int k = 0; k < 2; ++k) KT_HIP(hipMemsetAsync(t->wrkc_carry[k], 0, sizeof(float) * (size_t)t->cfg.cols * t->cfg.rows, t->ctx->stream));
    KT_TRY(kt_init_volume(t->ctx, t->tsdf, t->N));
    KT_TRY(kt_init_color_volume(t->ctx, t->color, t->N));
    KT_HIP(hipMemsetAsync(t->bricks, 0, kt_brick_count(t->N), t->ctx->stream));
    memset(t->stage_ms_sum, 0, sizeof(t->stage_ms_sum));
    memset(t->stage_n, 0, sizeof(t->stage_n));
    memset(t->stage_ms_last, 0, sizeof(t->stage_ms_last));
    t->last_U = t->last_S = 0;
    KT_HIP(hipStreamSynchronize(t->ctx->stream));
    return KT_OK;
}

}  // extern "C"

// ---- odometry set-up shared by ICP and RGB-D ------------------------------------------------------
static int odometry_begin(kt_tracker* t, const kt_level_k* first_k)
{
    kt_track_state* s = t->state_host;
    memset(s, 0, sizeof(*s));
    memcpy(s->Rprev, t->Rlast, sizeof(s->Rprev));
    memcpy(s->tprev, t->tlast, sizeof(s->tprev));
    memcpy(s->Rcurr, t->Rlast, sizeof(s->Rcurr));
    memcpy(s->tcurr, t->tlast, sizeof(s->tcurr));
    kt_mat33_inverse(s->Rprev, s->Rprev_inv);  // ICPOdometry.cpp:81
    for (int k = 0; k < 16; ++k) s->resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (first_k) kt_compute_krk(s->resultRt, *first_k, s->krkinv, s->kt);
    KT_HIP(hipMemcpyAsync(t->state_dev, s, sizeof(*s), hipMemcpyHostToDevice, t->ctx->stream));
    return KT_OK;
}

static int odometry_end(kt_tracker* t)
{
    return ev_end(t, ST_ODOMETRY);  // the pose stays on the device; complete_frame() reads it one frame later
}

static int fill_setup_args(kt_tracker* t, int mode, const float* R, const float* tv, kt_setup_args& a);

// ICPOdometry::getIncrementalTransformation, ICPOdometry.cpp:68-186
static int icp_odometry(kt_tracker* t, bool stepwise_only = false)
{
    int iters[KT_LEVELS] = {10, 5, 4, 0};
    if (t->cfg.fast_odometry) { iters[0] = 0; iters[1] = 10; iters[2] = 5; iters[3] = 0; }
    const float dist_thres = 0.10f;
    const float angle_thres = (float)sin(20.f * 3.14159254f / 180.f);  // ICPOdometry.h:35-36
    // ICPOdometry.cpp:70-85: the starting state travels in the arguments of the first iteration (no per-frame upload)
    kt_track_state init;
    memset(&init, 0, sizeof(init));
    memcpy(init.Rprev, t->Rlast, sizeof(init.Rprev));
    memcpy(init.tprev, t->tlast, sizeof(init.tprev));
    memcpy(init.Rcurr, t->Rlast, sizeof(init.Rcurr));
    memcpy(init.tcurr, t->tlast, sizeof(init.tcurr));
    kt_mat33_inverse(init.Rprev, init.Rprev_inv);  // ICPOdometry.cpp:81
    bool first = true;
    t->setup_fused = false;
    t->last_icp_levels = !stepwise_only && t->icp_levels && t->icp_demote == 0 && kt_live_trackers.load() == 1;
    if (!stepwise_only && t->icp_demote > 0) --t->icp_demote;
    if (t->last_icp_levels) {
        // ONE launch for the frame (round 6; round 5: one per level): the iterations of all levels hand the pose over inside the kernel
        // (kt_track.hip: kt_icp_level_kernel); KT_ICP_ONE_LAUNCH=0 restores the launch per level for A/B runs
        static const bool one_launch = []() { const char* e = getenv("KT_ICP_ONE_LAUNCH"); return !e || atoi(e) != 0; }();
        const float* vc[KT_LEVELS]; const float* nc[KT_LEVELS]; const float* vg[KT_LEVELS]; const float* ng[KT_LEVELS];
        kt_intr li[KT_LEVELS]; int cs[KT_LEVELS], rs[KT_LEVELS], its[KT_LEVELS], nl = 0;
        for (int l = KT_LEVELS - 1; l >= 0; --l) {
            if (iters[l] <= 0) continue;
            vc[nl] = t->vmaps_curr[l]; nc[nl] = t->nmaps_curr[l]; vg[nl] = t->vmaps_g_prev[l]; ng[nl] = t->nmaps_g_prev[l];
            li[nl] = lvl_intr(t->intr, l); cs[nl] = lvl_cols(t, l); rs[nl] = lvl_rows(t, l); its[nl] = iters[l];
            ++nl;
        }
        if (one_launch) {
            // ... and, optionally, the frame's set-up in that launch's epilogue (kt_setup.hpp; KT_ICP_FUSED_SETUP=1).  OFF by default: same poses and
            // volumes (tests/test_gpu_tracker.py), no faster where it was meant to be -- the serial odometry + set-up stage 0.153 ms either way: the
            // epilogue's checkpoint walks and z tables take what the kernel boundary took -- and SLOWER in the pipelined frame, 3790 against 3936 frames/s:
            // 256 full-CU workgroups that stay resident 5 us longer keep the waiting side-stream kernels out for as long, and those then run beside the
            // voxel kernel (0.15 against 0.19 of the roofline in the region; profiles/r06_experiments.md, call 19).
            // The epilogue's checkpoint blocks read the plan: the plan stream is joined in front of the launch instead of in front of the set-up.
            const char* fe = getenv("KT_ICP_FUSED_SETUP");
            const bool fuse_env = fe && atoi(fe) != 0;
            kt_setup_args su;
            t->setup_fused = fuse_env && !stepwise_only && nl > 0;
            if (t->setup_fused) {
                if (t->plan_sel >= 0 && hipEventQuery(t->plans[t->plan_sel].done) != hipSuccess) KT_HIP(hipStreamWaitEvent(t->ctx->stream, t->plans[t->plan_sel].done, 0));
                KT_TRY(fill_setup_args(t, 0, nullptr, nullptr, su));
                su.fused = 1;
            }
            if (nl) KT_TRY(kt_icp_levels_device(t->ctx, t->state_dev, nl, vc, nc, li, vg, ng, cs, rs, its, dist_thres, angle_thres, &init, 1, t->setup_fused ? &su : nullptr));
        } else {
            for (int k = 0; k < nl; ++k)
                KT_TRY(kt_icp_level_device(t->ctx, t->state_dev, vc[k], nc[k], &li[k], vg[k], ng[k], cs[k], rs[k], dist_thres, angle_thres, &init, k == 0 ? 1 : 0, its[k]));
        }
        return odometry_end(t);
    }
    for (int l = KT_LEVELS - 1; l >= 0; --l) {
        const kt_intr li = lvl_intr(t->intr, l);
        for (int it = 0; it < iters[l]; ++it) {
            KT_TRY(kt_icp_step_device(t->ctx, t->state_dev, t->vmaps_curr[l], t->nmaps_curr[l], &li, t->vmaps_g_prev[l],
                                      t->nmaps_g_prev[l], lvl_cols(t, l), lvl_rows(t, l), dist_thres, angle_thres, KT_MODE_ICP_SOLVE,
                                      first ? &init : nullptr));
            first = false;
        }
    }
    return odometry_end(t);
}

// RGBDOdometry::getIncrementalTransformation, RGBDOdometry.cpp:165-393
static int rgbd_odometry(kt_tracker* t, int set, int last_set, bool stepwise_only = false)
{
    int iters[KT_LEVELS];
    if (!t->cfg.use_rgbd_icp) {
        iters[0] = 10; iters[1] = 7; iters[2] = 7; iters[3] = 7;
        if (t->cfg.fast_odometry) { iters[0] = 0; iters[1] = 10; iters[2] = 7; iters[3] = 0; }
    } else {
        iters[0] = 10; iters[1] = 5; iters[2] = 4; iters[3] = 0;
        if (t->cfg.fast_odometry) { iters[0] = 0; iters[1] = 10; iters[2] = 7; iters[3] = 0; }
    }
    const double SOBEL_SCALE = 1.0 / pow(2.0, 3), MAX_DEPTH_DELTA = 0.07;
    const float dist_thres = 0.10f;
    const float angle_thres = (float)sin(20.f * 3.14159254f / 180.f);
    // populateRGBDData, the derivative images and the point clouds were built with the frame sets (build_frame_set)
    KT_ARG(last_set >= 0);
    const FrameSet& next = t->sets[set];
    const FrameSet& last = t->sets[last_set];

    const double ifx = t->intr.fx, ify = t->intr.fy, icx = t->intr.cx, icy = t->intr.cy;  // RGBDOdometry.cpp:72-75
    kt_level_k lk[KT_LEVELS];
    for (int l = 0; l < KT_LEVELS; ++l) {
        const int div = 1 << l;
        lk[l].fx = ifx / div; lk[l].fy = ify / div; lk[l].cx = icx / div; lk[l].cy = icy / div;
    }
    // flatten the (level, iteration) schedule so each solve epilogue knows the NEXT iteration's level
    int sched[64], ns = 0;
    for (int l = KT_LEVELS - 1; l >= 0; --l)
        for (int j = 0; j < iters[l]; ++j) sched[ns++] = l;
    KT_TRY(odometry_begin(t, ns ? &lk[sched[0]] : nullptr));
    // -ri: the iterations of a pyramid level in ONE launch (round 6, kt_track.hip: kt_joint_level_kernel), under the conditions of the ICP chain's
    // level form (alone in the process, not demoted by a recent fallback, the grid fits the device).  OFF by default: bit-equal to the two
    // launches per iteration (tests/test_gpu_tracker.py, config 3), but measured SLOWER on the crab-walk -- 1980 against 2100-2300 frames/s
    // (profiles/r06_experiments.md, call 10): the odometry stage itself is no shorter (0.303 against 0.301 ms: what two kernel boundaries cost,
    // the DataTerm round trip through the L2 and 186 spilled scalar registers cost back), and a resident 1024-thread launch keeps the RGB-D
    // read-ahead (160 us per frame) out of every compute unit for the whole odometry.  KT_RI_LEVELS=1 / kt_debug_ri_levels(1) select it.
    const bool ri_levels_env = kt_ri_levels_selected();
    const bool ri_levels = t->cfg.use_rgbd_icp && !stepwise_only && ri_levels_env && t->icp_levels && t->icp_demote == 0 && kt_live_trackers.load() == 1;
    if (t->cfg.use_rgbd_icp && !stepwise_only && t->icp_demote > 0) --t->icp_demote;
    t->last_icp_levels = ri_levels;
    if (ri_levels) {
        for (int q = 0; q < ns;) {
            const int l = sched[q];
            int cnt = 0;
            while (q + cnt < ns && sched[q + cnt] == l) ++cnt;
            const kt_intr li = lvl_intr(t->intr, l);
            const kt_level_k* k_next = &lk[q + cnt < ns ? sched[q + cnt] : l];
            KT_TRY(kt_joint_level_device(t->ctx, t->state_dev, t->vmaps_curr[l], t->nmaps_curr[l], &li, t->vmaps_g_prev[l], t->nmaps_g_prev[l], dist_thres, angle_thres,
                                         t->corres[l], last.cloud[l], next.dIdx[l], next.dIdy[l], (float)SOBEL_SCALE, rgbd_min_scale(l), last.depth_m[l], next.depth_m[l],
                                         last.image[l], next.image[l], (float)MAX_DEPTH_DELTA, next.cand[l], lvl_cols(t, l), lvl_rows(t, l), cnt, &lk[l], k_next));
            q += cnt;
        }
        return odometry_end(t);
    }
    for (int q = 0; q < ns; ++q) {
        const int l = sched[q];
        const int cols = lvl_cols(t, l), rows = lvl_rows(t, l);
        const int first_at_level = q == 0 || sched[q - 1] != l;   // writes every DataTerm; later iterations only the candidates
        KT_TRY(kt_rgb_residual_device(t->ctx, t->state_dev, rgbd_min_scale(l), next.dIdx[l], next.dIdy[l], last.depth_m[l], next.depth_m[l],
                                      last.image[l], next.image[l], cols, rows, t->corres[l], (float)MAX_DEPTH_DELTA, next.cand[l],
                                      first_at_level));
        const kt_intr li = lvl_intr(t->intr, l);
        const kt_level_k* nk = &lk[q + 1 < ns ? sched[q + 1] : l];
        if (t->cfg.use_rgbd_icp)   // ICP sums + RGB-D sums + joint solve in one launch
            KT_TRY(kt_joint_step_device(t->ctx, t->state_dev, t->vmaps_curr[l], t->nmaps_curr[l], &li, t->vmaps_g_prev[l], t->nmaps_g_prev[l],
                                        dist_thres, angle_thres, t->corres[l], last.cloud[l], next.dIdx[l], next.dIdy[l], (float)SOBEL_SCALE,
                                        cols, rows, nk));
        else
            KT_TRY(kt_rgb_step_device(t->ctx, t->state_dev, t->corres[l], last.cloud[l], li.fx, li.fy, next.dIdx[l], next.dIdy[l],
                                      (float)SOBEL_SCALE, cols, rows, KT_MODE_RGB_SOLVE, nk));
    }
    // swap last/next (:377-381) = the frame sets rotate; the > 0.3 m jump guard (:383-387) runs in kt_frame_setup_kernel
    return odometry_end(t);
}

// ---- frame set-up on the device ---------------------------------------------------------------------------------------
// Runs after the last odometry iteration (csrc/kt_setup.hpp: what it does, shared with the epilogue of kt_icp_level_kernel).  A frame that
// must shift the volume first is parked (skip = 1) and redone by the host's shift path in complete_frame().
// Workgroup 0 (two waves) is the set-up proper; workgroups 1.. are the carry and checkpoint blocks.
__global__ __launch_bounds__(256) void kt_frame_setup_kernel(const kt_setup_args a)
{
    if (blockIdx.x > 0) {
        float R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tv[3] = {0, 0, 0};
        if ((int)blockIdx.x > a.carry_groups) kt_setup_final_pose(a, R, tv);   // (the checkpoint blocks of a planned frame: mode 0)
        kt_setup_side_block(a, (int)blockIdx.x, (int)threadIdx.x, R, tv);
        return;
    }
    if (threadIdx.x >= 128 || a.mode == 2) return;
    float R[9], tv[3];
    if (a.mode == 0) kt_setup_final_pose(a, R, tv);
    else {
        for (int k = 0; k < 9; ++k) R[k] = a.R[k];
        for (int k = 0; k < 3; ++k) tv[k] = a.t[k];
    }
    kt_setup_block0(a, R, tv, a.mode == 0 ? a.st->handoff_timeout : 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

// the arguments of the frame's set-up (kt_setup.hpp) as the tracker stands: for kt_frame_setup_kernel, or for the epilogue of the odometry launch
static int fill_setup_args(kt_tracker* t, int mode, const float* R, const float* tv, kt_setup_args& a)
{
    KT_TRY(kt_integrate_tables(t->ctx, t->cfg.cols, t->cfg.rows, t->N, &t->vgz_dev, &t->zs_dev));  // may have been regrown by another user of the context
    a.st = t->state_dev; a.fp = t->fp_dev; a.vgz = t->vgz_dev; a.zs = t->zs_dev; a.N = t->N;
    a.mirror = t->mirror; a.seq = t->frame_seq;
    a.cell_z = t->volume_size[2] / t->N;
    a.mode = mode;
    a.rgbd_guard = (t->cfg.use_rgbd || t->cfg.use_rgbd_icp) ? 1 : 0;
    for (int k = 0; k < 9; ++k) a.R[k] = R ? R[k] : 0.f;
    for (int k = 0; k < 3; ++k) a.t[k] = tv ? tv[k] : 0.f;
    for (int k = 0; k < 3; ++k) { a.basis[k] = t->volume_basis[k]; a.voxel[k] = t->voxel_size[k]; }
    a.thresh = t->parked ? INT_MAX : t->cfg.voxel_shift;
    a.rec = (kt_pixrec*)t->sets[t->out_set].rec;
    a.carry_cur = t->wrkc_carry[t->carry_sel];
    a.carry_next = t->wrkc_carry[t->carry_sel ^ 1];
    a.npix = t->cfg.cols * t->cfg.rows;
    a.carry_groups = t->cfg.disable_color_angle ? 0 : (a.npix + 1023) / 1024;   // without the angle weight wrkc is 2 everywhere
    a.plan_wrange = nullptr; a.plan_walk0 = nullptr;
    a.walk_groups = 0;
    a.fused = 0;
    a.wx = a.wy = a.wcx = a.wcy = a.XG = a.YG = 0; a.cell_x = a.cell_y = a.fx = a.fy = 0.0f; a.plan_theta = a.plan_tau = 0.0f;
    memset(a.plan_R, 0, sizeof(a.plan_R)); memset(a.plan_t, 0, sizeof(a.plan_t));
    if (mode == 0 && t->plan_sel >= 0) {
        const kt_tracker::PlanSlot& pl = t->plans[t->plan_sel];
        a.plan_wrange = pl.plan.wrange; a.plan_walk0 = pl.plan.walk0;
        memcpy(a.plan_R, pl.R, sizeof(a.plan_R)); memcpy(a.plan_t, pl.t, sizeof(a.plan_t));
        a.plan_theta = pl.theta; a.plan_tau = pl.tau;
        kt_tsdf_plan_shape(t->cfg.cols, t->cfg.rows, t->N, &a.wcx, &a.wcy, &a.XG, &a.YG);
        a.wx = t->v_wrap_copy[0] % t->N; a.wy = t->v_wrap_copy[1] % t->N;
        a.cell_x = t->volume_size[0] / t->N; a.cell_y = t->volume_size[1] / t->N;
        a.fx = t->intr.fx; a.fy = t->intr.fy;
        a.walk_groups = (a.XG * a.YG + 3) / 4;
    }
    return KT_OK;
}

static int launch_setup(kt_tracker* t, int mode, const float* R, const float* tv)
{
    kt_setup_args a;
    KT_TRY(fill_setup_args(t, mode, R, tv, a));
    hipLaunchKernelGGL(kt_frame_setup_kernel, dim3(1 + a.carry_groups + a.walk_groups), dim3(256), 0, t->ctx->stream, a);
    KT_LAUNCH_CHECK();
    return KT_OK;
}

// [H] integrate (:864-876) + [I] raycast (:880-890) + [J] predicted-map pyramid (:892-899, fused into the raycast epilogue), with
// the pose taken from fp_dev; wrap = the tracker's current v_wrap_copy
static int enqueue_fusion(kt_tracker* t, int set, const uint16_t* depth_raw, const uint8_t* colors, const kt_tsdf_plan* plan)
{
    kt_ctx* c = t->ctx;
    const int cols = t->cfg.cols, rows = t->cfg.rows, N = t->N;
    const bool icp = !t->has_trajectory && !(t->cfg.use_rgbd || t->cfg.use_rgbd_icp);
    const kt_mat33 dummy_R = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    const float dummy_t[3] = {0, 0, 0};
    if (t->counting) {
        KT_HIP(hipMemsetAsync(t->upd_dev, 0, 16 * sizeof(unsigned int), c->stream));
        KT_HIP(hipMemsetAsync(t->steps_dev, 0, 4 * sizeof(unsigned long long), c->stream));
    }
    KT_TRY(ev_begin(t, ST_INTEGRATE));
    tsdf23_hook_arm(t);
    KT_TRY(kt_integrate_tsdf_impl(c, depth_raw, cols, rows, &t->intr, t->volume_size, &dummy_R, dummy_t, t->tranc_dist, t->tsdf,
                                  t->sets[set].scaled, t->v_wrap_copy, t->color, colors, t->sets[set].nmaps[0], !t->cfg.disable_color_angle, N,
                                  t->counting ? t->upd_dev : nullptr, t->sets[set].rec, t->fp_dev, t->bricks, t->sets[set].dpmax, plan));
    KT_TRY(ev_end(t, ST_INTEGRATE));
    if (t->side_gate) { KT_HIP(hipEventRecord(t->gate_ev, c->stream)); t->gate_armed = true; }   // the voxel kernel has drained: the side streams may run
    KT_TRY(ev_begin(t, ST_RAYCAST));
    const bool pyr = icp || t->cfg.use_rgbd_icp;
    float* vp[3] = {t->vmaps_g_prev[1], t->vmaps_g_prev[2], t->vmaps_g_prev[3]};
    float* np_[3] = {t->nmaps_g_prev[1], t->nmaps_g_prev[2], t->nmaps_g_prev[3]};
    KT_TRY(kt_raycast_impl(c, &t->intr, &dummy_R, dummy_t, t->tranc_dist, t->volume_size, t->tsdf, t->vmaps_g_prev[0], t->nmaps_g_prev[0], cols, rows,
                           t->v_wrap_copy, t->vmap_curr_color, t->color, N, t->counting ? t->steps_dev : nullptr, pyr ? vp : nullptr,
                           pyr ? np_ : nullptr, t->fp_dev, t->bricks));
    KT_TRY(ev_end(t, ST_RAYCAST));
    return KT_OK;
}

// wait for the slice downloads in flight (every reader of slices[i].pts goes through here)
static int join_slice_jobs(kt_tracker* t)
{
    int rc = KT_OK;
    for (int b = 0; b < 2; ++b) {
        kt_tracker::SliceJob& j = t->jobs[b];
        if (!j.active) continue;
        { std::unique_lock<std::mutex> lk(t->wmu); t->wdone.wait(lk, [&j]() { return j.done; }); }
        j.active = false;
        if (j.status != hipSuccess) { kt_set_error("slice download: %s", hipGetErrorString(j.status)); rc = KT_ERR_HIP; }
    }
    return rc;
}

// fetchCloud + download (TSDFVolume.cpp:135-172, KintinuousTracker.cpp:1164-1166).  Nothing here waits for the GPU: the kernel, the copy
// of its count and an event are enqueued, and a helper thread moves the points into the slice once the event has fired.
static int fetch_slice(kt_tracker* t, const int lo[3], const int hi[3], int dim)
{
    kt_ctx* c = t->ctx;
    const int b = t->cloud_next;
    t->cloud_next ^= 1;
    kt_tracker::SliceJob& j = t->jobs[b];
    if (j.active) {   // the download that used this buffer two slices ago
        { std::unique_lock<std::mutex> lk(t->wmu); t->wdone.wait(lk, [&j]() { return j.done; }); }
        j.active = false;
        if (j.status != hipSuccess) { kt_set_error("slice download: %s", hipGetErrorString(j.status)); return KT_ERR_HIP; }
    }
    const bool stage = t->slice_stage;
    if (!stage) {
        KT_TRY(kt_extract_cloud_slice_async(c, t->tsdf, t->volume_size, t->cloud_host[b], t->cloud_cap, t->v_wrap_copy, t->color, lo[0], hi[0],
                                            lo[1], hi[1], lo[2], hi[2], 1, t->voxel_wrap, t->N, &c->counters[1]));
        KT_HIP(hipMemcpyAsync(t->cloud_count_host[b], &c->counters[1], sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        KT_HIP(hipEventRecord(t->cloud_ev[b], c->stream));
    } else {
        // the slab stays on the device; everything behind the extraction kernel runs on the slice stage's own stream
        hipStream_t ss = (hipStream_t)kt_slice_ws_stream(t->slice_ws);
        KT_TRY(kt_extract_cloud_slice_async(c, t->tsdf, t->volume_size, t->cloud_dev[b], t->cloud_cap, t->v_wrap_copy, t->color, lo[0], hi[0],
                                            lo[1], hi[1], lo[2], hi[2], 1, t->voxel_wrap, t->N, t->cloud_count_dev[b]));
        KT_HIP(hipEventRecord(t->extracted[b], c->stream));
        KT_HIP(hipStreamWaitEvent(ss, t->extracted[b], 0));
        KT_TRY(kt_copy_counted(ss, t->cloud_dev[b], t->cloud_host[b], t->cloud_count_dev[b], (unsigned int)t->cloud_cap, (int)sizeof(kt_point_xyzrgb), t->cloud_count_host[b]));
        const float leaf = fmaxf(t->voxel_size[0], fmaxf(t->voxel_size[1], t->voxel_size[2]));   // CloudSliceProcessor.cpp:124-130
        KT_TRY(kt_slice_process_device(t->slice_ws, t->cloud_dev[b], t->cloud_count_dev[b], t->cloud_cap, t->slice_cull, leaf, t->slice_k));
        KT_TRY(kt_copy_counted(ss, kt_slice_ws_output(t->slice_ws), t->proc_host[b], kt_slice_ws_leaves_dev(t->slice_ws), (unsigned int)t->cloud_cap,
                               (int)sizeof(kt_point_xyzrgbnormal), t->proc_count_host[b]));
        KT_HIP(hipEventRecord(t->cloud_ev[b], ss));
    }
    t->slices.emplace_back();
    Slice& s = t->slices.back();
    s.dim = dim;
    s.pr_id = -1;
    memcpy(s.R, t->Rlast, sizeof(s.R));                          // rmats_.back(), currentGlobalCamera, current_utime:
    memcpy(s.cam, t->current_global_camera, sizeof(s.cam));      // KintinuousTracker.cpp:1186-1191
    s.ts = t->current_ts;
    j.slice = t->slices.size() - 1;
    j.status = hipSuccess;
    j.active = true;
    Slice* dst = &s;   // references into a deque survive push_back
    const kt_point_xyzrgb* src = t->cloud_host[b];
    const unsigned int* cnt = t->cloud_count_host[b];
    const size_t cap = t->cloud_cap;
    hipEvent_t ev = t->cloud_ev[b];
    hipError_t* status = &j.status;
    const kt_point_xyzrgbnormal* psrc = stage ? t->proc_host[b] : nullptr;
    const unsigned int* pcnt = stage ? t->proc_count_host[b] : nullptr;
    bool* done = &j.done;
    std::mutex* mu = &t->wmu;
    j.done = false;
    {
        std::lock_guard<std::mutex> lk(t->wmu);
        t->wq.emplace_back([dst, src, cnt, cap, ev, status, psrc, pcnt, done, mu]() {
            const hipError_t e = hipEventSynchronize(ev);   // the kernel's stores to host memory and the count are complete
            if (e != hipSuccess) *status = e;
            else {
                size_t n = (size_t)*cnt;
                if (n > cap) n = cap;
                dst->pts.assign(src, src + n);
                if (psrc) {
                    size_t m = (size_t)*pcnt;
                    if (m > cap) m = cap;
                    dst->processed.assign(psrc, psrc + m);
                    dst->has_processed = true;
                }
            }
            std::lock_guard<std::mutex> lk2(*mu);
            *done = true;
        });
    }
    t->wcv.notify_one();
    return KT_OK;
}

static int read_counts(kt_tracker* t)
{
    if (!t->counting) return KT_OK;
    kt_ctx* c = t->ctx;
    unsigned int u = 0;
    unsigned long long sdev = 0;
    KT_HIP(hipMemcpyAsync(&u, t->upd_dev, sizeof(u), hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipMemcpyAsync(&sdev, t->steps_dev, sizeof(sdev), hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    t->last_U = u;
    t->last_S = sdev;
    return KT_OK;
}

// GroundTruthOdometry::getIncrementalTransformation (GroundTruthOdometry.cpp:42-74): the pose of the frame stamped `timestamp` from
// the loaded trajectory, composed in the volume's frame.  Every product is float, in Eigen's evaluation order: delta =
// Ta^-1 * Tb as Isometry3f (3x3 products, translation = linear * t + t), then currentTsdf * M^-1 * delta * M left to right as
// 4x4 products.  M permutes and negates axes, so the first and last product are exact column moves; the middle one rounds, with
// its sums running over the moved columns.
static void ground_truth_pose(kt_tracker* t, uint64_t timestamp, float Rcurr[9], float tcurr[3])
{
    memcpy(Rcurr, t->Rlast, 9 * sizeof(float));
    memcpy(tcurr, t->tlast, 3 * sizeof(float));
    if (t->gt_utime == 0 || t->trajectory.empty()) return;   // :50 -- a previous stamp of 0 reads as "no previous frame"
    const std::array<float, 12>& A = t->trajectory[(int)(uint32_t)t->gt_utime];   // operator[], as the reference
    const std::array<float, 12>& B = t->trajectory[(int)(uint32_t)timestamp];
    kt_host_ground_truth_pose(A.data(), B.data(), t->Rlast, t->tlast, Rcurr, tcurr);
}

// KintinuousTracker::addToPlaceRecognition :917-958: the sample carries lastPlaceRecognitionTrans / Rot (set by the caller just before)
static int add_pr_sample(kt_tracker* t)
{
    PrSample ps;
    ps.utime = t->current_ts;
    memcpy(ps.trans, t->pr_trans, sizeof(ps.trans));
    memcpy(ps.rot, t->pr_rot, sizeof(ps.rot));
    ps.pose_index = (int)t->poses.size();   // the dense pose of the frame being processed is pushed after this
    t->pr_samples.push_back(ps);
    return (int)t->pr_samples.size() - 1;
}

// Host half of a frame once its pose is known: the pose bookkeeping of KintinuousTracker.cpp:574-595, 903-909 and, when the volume
// has to shift (:627-833), the shift.  speculated = the fusion kernels were already enqueued against the device-side shift decision
// (they ran unless the device parked them); otherwise (pose supplied by the host) they are enqueued here, after any shift.
static int finish_pose(kt_tracker* t, float Rcurr[9], float tcurr[3], bool speculated)
{
    kt_ctx* c = t->ctx;
    const int N = t->N;
    t->current_ts = t->out_ts;

    // [D] rmats_/tvecs_ push, currentGlobalCamera :574-595
    memcpy(t->Rlast, Rcurr, 9 * sizeof(float));
    memcpy(t->tlast, tcurr, 3 * sizeof(float));
    compute_global_camera(t, tcurr);
    if (t->cfg.dynamic_cube)  // :597-600; while parked its threshold is VOLUME_X (x 3 with -sm), :403
        kt_host_reposition_cube(Rcurr, t->tlast, t->cfg.volume_size, t->voxel_size,
                                t->parked ? (t->cfg.static_mode ? t->N * 3 : t->N) : t->cfg.voxel_shift, t->volume_basis);

    // place-recognition tap :601-624: sample now if the camera has moved enough since the last sample, else at the next shift
    bool shift_send = false;
    int is_loop = 0;
    if (t->cfg.place_recognition) {
        const float place_recognition_movement = 0.15f;   // KintinuousTracker.cpp:76
        if (kt_host_place_recognition_movement(Rcurr, t->current_global_camera, t->pr_rot, t->pr_trans) >= place_recognition_movement) {
            memcpy(t->pr_rot, Rcurr, sizeof(t->pr_rot));
            memcpy(t->pr_trans, t->current_global_camera, sizeof(t->pr_trans));
            add_pr_sample(t);
            is_loop = 1;
        } else {
            shift_send = true;
        }
    }

    // [F] shift decision :627-667 and the three axis blocks :669-833
    float current_translation[3];
    for (int k = 0; k < 3; ++k) current_translation[k] = t->tlast[k] - t->volume_basis[k];
    const int thresh = t->out_thresh;
    int vt[3];
    bool need_shift = false;
    for (int k = 0; k < 3; ++k) {
        vt[k] = voxel_trans(current_translation[k], t->voxel_size[k], thresh);
        need_shift = need_shift || vt[k] >= thresh || vt[k] <= -thresh;
    }
    if (speculated && need_shift != (t->mirror->skip == 1)) {
        kt_set_error("tracker: host and device disagree on the shift decision");
        return KT_ERR_STATE;
    }
    // skip == 2: the pose fell outside the margins the frame's task plan was made with -- the device parked the fusion kernels, the
    // frame is fused below through the in-stream pre-pass
    const bool plan_missed = speculated && t->mirror->skip == 2;
    if (speculated && t->plan_sel >= 0) {
        float dr = 0.0f, dt = 0.0f;
        const kt_tracker::PlanSlot& pl = t->plans[t->plan_sel];
        for (int k = 0; k < 9; ++k) dr += (Rcurr[k] - pl.R[k]) * (Rcurr[k] - pl.R[k]);
        for (int k = 0; k < 3; ++k) dt += (tcurr[k] - pl.t[k]) * (tcurr[k] - pl.t[k]);
        t->pred_err_r = sqrtf(dr) * 0.70710678f;   // |R - R^|_F ~ sqrt(2) angle
        t->pred_err_t = sqrtf(dt);
        if (plan_missed) ++t->plan_misses; else if (!need_shift) ++t->plan_hits;
    }
    t->plan_sel = -1;   // whatever is enqueued from here on for this frame runs its own pre-pass
    if (need_shift) {
        const int ov = t->cfg.overlap;
        KT_TRY(ev_begin(t, ST_SHIFT));
        for (int axis = 0; axis < 3; ++axis) {
            v_wrap_copy_update(t);
            bool cycled = false;
            int lo[3] = {0, 0, 0}, hi[3] = {N, N, N};
            int dim = 0;
            if (vt[axis] >= thresh) {
                lo[axis] = 0; hi[axis] = vt[axis] + 1 + ov;
                dim = axis * 2;  // XPlus / YPlus / ZPlus, CloudSlice.h:33-36
                KT_TRY(fetch_slice(t, lo, hi, dim));
                KT_TRY(kt_clear_volume(c, t->tsdf, 2, N, axis, 0, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]));
                KT_TRY(kt_clear_volume(c, t->color, 4, N, axis, 0, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]));
                cycled = true;
            } else if (vt[axis] <= -thresh) {
                if (axis == 2) { lo[2] = N + (vt[2] - ov) - 1; hi[2] = N - 1; }  // z-minus off by one :805
                else { lo[axis] = N + (vt[axis] - ov); hi[axis] = N; }
                dim = axis * 2 + 1;
                KT_TRY(fetch_slice(t, lo, hi, dim));
                KT_TRY(kt_clear_volume(c, t->tsdf, 2, N, axis, 1, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]));
                KT_TRY(kt_clear_volume(c, t->color, 4, N, axis, 1, t->voxel_wrap[axis], t->voxel_wrap[axis] + vt[axis]));
                cycled = true;
            }
            if (cycled) {
                if (shift_send) {   // :706-717, :762-771, :816-825: the slab leaving the volume takes a sample with it
                    memcpy(t->pr_rot, Rcurr, sizeof(t->pr_rot));
                    memcpy(t->pr_trans, t->current_global_camera, sizeof(t->pr_trans));
                    t->slices.back().pr_id = add_pr_sample(t);
                    is_loop = 1;
                    shift_send = false;
                }
                // mutexOutCloudBuffer :1156-1208
                const float shift = t->voxel_size[axis] * (float)vt[axis];
                t->tlast[axis] -= shift;
                t->voxel_wrap[axis] += vt[axis];
                tcurr[axis] -= shift;
            }
        }
        v_wrap_copy_update(t);
        KT_TRY(ev_end(t, ST_SHIFT));
    }
    if (need_shift || !speculated || plan_missed) {
        // the fusion the device parked (or that was never enqueued), with the final pose and wrap
        v_wrap_copy_update(t);
        KT_TRY(launch_setup(t, 1, Rcurr, tcurr));
        KT_TRY(enqueue_fusion(t, t->out_set, t->out_depth, t->out_rgb, nullptr));
    }
    v_wrap_copy_update(t);
    ++t->global_time;
    push_pose(t, t->out_ts, Rcurr, is_loop);  // [K] :903-909
    // pose history for the next prediction (rotation and the shift-invariant global camera)
    memcpy(t->hist_R[0], t->hist_R[1], sizeof(t->hist_R[0])); memcpy(t->hist_gc[0], t->hist_gc[1], sizeof(t->hist_gc[0]));
    memcpy(t->hist_R[1], Rcurr, sizeof(t->hist_R[1])); memcpy(t->hist_gc[1], t->current_global_camera, sizeof(t->hist_gc[1]));
    if (t->hist_n < 2) ++t->hist_n;
    return KT_OK;
}

// Planning ahead.  Called in process_frame(f), before f's odometry is enqueued, for the frame f + 1 that has been read ahead: the host
// knows the poses of frames f - 1 and f - 2, predicts the pose of f + 1 by repeating the last motion increment twice
// (body-frame rotation increment, global-camera translation increment), and enqueues the voxel kernel's pre-pass for that prediction
// on plan_stream behind the frame's read-ahead -- where it runs next to the fusion kernels of frame f - 1.  Margins: three times the
// error of the previous prediction on top of a floor, within caps.  The set-up kernel of frame f + 1 checks its pose against them;
// a volume shift in between invalidates the plan (its storage wrap no longer matches).
// A frame set that leaves the pending list WITHOUT being processed (a skipped or dropped read-ahead) takes its plan with it: the slot is
// invalidated, and whatever recycles the set -- the main stream or the read-ahead stream -- first waits for the plan's kernels, which read
// the set's pixel records and tile maxima on plan_stream.
static int drop_plans_of_set(kt_tracker* t, int set)
{
    for (int k = 0; k < 3; ++k) {
        kt_tracker::PlanSlot& pl = t->plans[k];
        if (pl.ordinal < 0 || pl.set != set) continue;
        KT_HIP(hipStreamWaitEvent(t->ctx->stream, pl.done, 0));
        KT_HIP(hipStreamWaitEvent(t->pre_stream, pl.done, 0));
        pl.ordinal = -1; pl.set = -1;
    }
    return KT_OK;
}

static float plan_rand(kt_tracker* t)   // uniform in (-1, 1), deterministic per tracker
{
    t->plan_rng = t->plan_rng * 1664525u + 1013904223u;
    return (float)((t->plan_rng >> 8) & 0xffffffu) * (2.0f / 16777216.0f) - 1.0f;
}
static void plan_rand_unit(kt_tracker* t, float v[3])
{
    for (;;) {
        v[0] = plan_rand(t); v[1] = plan_rand(t); v[2] = plan_rand(t);
        const float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        if (n2 > 0.01f && n2 <= 1.0f) { const float s = 1.0f / sqrtf(n2); v[0] *= s; v[1] *= s; v[2] *= s; return; }
    }
}

static int plan_ahead(kt_tracker* t, const Pending& next, long long ordinal)
{
    const int set = next.set;
    kt_tracker::PlanSlot& pl = t->plans[ordinal % 3];
    pl.ordinal = -1; pl.set = -1;
    float D[9], Rp[9], tp[3];
    for (int i = 0; i < 3; ++i)       // D = R(f-2)^T R(f-1)
        for (int j = 0; j < 3; ++j) D[i * 3 + j] = t->hist_R[0][0 * 3 + i] * t->hist_R[1][0 * 3 + j] + t->hist_R[0][1 * 3 + i] * t->hist_R[1][1 * 3 + j] + t->hist_R[0][2 * 3 + i] * t->hist_R[1][2 * 3 + j];
    for (int i = 0; i < 3; ++i)       // R^ = R(f-1) D
        for (int j = 0; j < 3; ++j) Rp[i * 3 + j] = t->Rlast[i * 3 + 0] * D[0 * 3 + j] + t->Rlast[i * 3 + 1] * D[1 * 3 + j] + t->Rlast[i * 3 + 2] * D[2 * 3 + j];
    float step = 0.0f, turn = 0.0f;
    for (int k = 0; k < 3; ++k) {
        const float d = t->hist_gc[1][k] - t->hist_gc[0][k];
        tp[k] = t->tlast[k] + d;
        step += d * d;
    }
    for (int k = 0; k < 9; ++k) turn += (D[k] - ((k % 4 == 0) ? 1.0f : 0.0f)) * (D[k] - ((k % 4 == 0) ? 1.0f : 0.0f));
    step = sqrtf(step); turn = sqrtf(turn) * 0.70710678f;
    if (t->plan_hits + t->plan_misses == 0) { t->pred_err_t = 0.5f * step; t->pred_err_r = 0.5f * turn; }   // nothing observed yet
    pl.tau = t->plan_margin_scale * fminf(0.020f, 0.0015f + 3.0f * t->pred_err_t);
    pl.theta = t->plan_margin_scale * fminf(0.02f, 3.0e-4f + 3.0f * t->pred_err_r);
    // ---- test hooks: where a plan can actually fail is a pose that lands just inside its margins
    if (t->plan_theta_fixed > 0.0f) pl.theta = t->plan_theta_fixed;
    if (t->plan_tau_fixed > 0.0f) pl.tau = t->plan_tau_fixed;
    if ((size_t)(ordinal + 1) * 12 <= t->plan_truth.size()) {   // the pose frame `ordinal` arrived at in an identical earlier run
        memcpy(Rp, &t->plan_truth[(size_t)ordinal * 12], sizeof(Rp));
        memcpy(tp, &t->plan_truth[(size_t)ordinal * 12 + 9], sizeof(tp));
    }
    if (t->plan_perturb) {   // prediction := prediction (+) a rotation of fr * theta about a random axis, + ft * tau along a random direction
        float ax[3], dir[3], K[9], Rd[9], Rn[9];
        plan_rand_unit(t, ax); plan_rand_unit(t, dir);
        const float phi = t->plan_fr * pl.theta, sn = sinf(phi), cs = cosf(phi);
        const float Kx[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) K[i * 3 + j] = Kx[i * 3 + 0] * Kx[0 * 3 + j] + Kx[i * 3 + 1] * Kx[1 * 3 + j] + Kx[i * 3 + 2] * Kx[2 * 3 + j];
        for (int k = 0; k < 9; ++k) Rd[k] = ((k % 4 == 0) ? 1.0f : 0.0f) + sn * Kx[k] + (1.0f - cs) * K[k];   // Rodrigues
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = Rp[i * 3 + 0] * Rd[0 * 3 + j] + Rp[i * 3 + 1] * Rd[1 * 3 + j] + Rp[i * 3 + 2] * Rd[2 * 3 + j];
        memcpy(Rp, Rn, sizeof(Rp));
        for (int k = 0; k < 3; ++k) tp[k] += dir[k] * t->plan_ft * pl.tau;
    }
    kt_mat33 Rinv;
    kt_mat33_inverse(Rp, Rinv.m);
    v_wrap_copy_update(t);
    KT_HIP(hipStreamWaitEvent(t->plan_stream, t->sets[set].ready, 0));
    const int made = kt_integrate_plan(t->plan_stream, &pl.plan, t->sets[set].rec, t->sets[set].dpmax, t->cfg.cols, t->cfg.rows, &t->intr, t->volume_size, &Rinv, tp,
                                       t->tranc_dist, t->v_wrap_copy, t->N, pl.theta, pl.tau);
    if (made == KT_NO_PLAN) return KT_OK;   // margins too wide for this camera: planning is an optimisation, the frame takes the in-stream pre-pass
    KT_TRY(made);
    KT_HIP(hipEventRecord(pl.done, t->plan_stream));
    memcpy(pl.R, Rp, sizeof(pl.R)); memcpy(pl.t, tp, sizeof(pl.t));
    memcpy(pl.wrap, t->v_wrap_copy, sizeof(pl.wrap));
    pl.set = set; pl.depth = next.depth; pl.rgb = next.rgb;
    pl.ordinal = ordinal;
    return KT_OK;
}

// plan `frame` as the frame with this ordinal, if planning applies; sets plan_sel when the frame is the one being handed over (ordinal ==
// frames_started - 1 is decided by the caller: process_frame passes its own ordinal and takes plan_sel from the slot)
static int plan_frame(kt_tracker* t, const Pending& frame, long long ordinal)
{
    // (-d repositions the cube once the pose is known, a ground-truth trajectory supplies the pose on the host: nothing to plan for)
    if (!(t->plan_enabled && t->hist_n >= 2 && !t->cfg.dynamic_cube && !t->has_trajectory)) return KT_OK;
    // on a gated (dense) view the plan stream is held like the read-ahead: not beside the voxel kernel that is running now
    if (t->side_gate && t->gate_armed) KT_HIP(hipStreamWaitEvent(t->plan_stream, t->gate_ev, 0));
    v_wrap_copy_update(t);
    KT_TRY(plan_ahead(t, frame, ordinal));
    if (t->plans[ordinal % 3].ordinal == ordinal && ordinal == t->frames_started - 1) t->plan_sel = (int)(ordinal % 3);
    return KT_OK;
}

// Host half of the frame enqueued by the last kt_tracker_process_frame call: wait for its pose (the copy-stream event, NOT the
// fusion kernels), do the pose bookkeeping of KintinuousTracker.cpp:574-595, 903-909 and, when the volume has to shift
// (:627-833), run the shift and redo the fusion the device parked.  Called at the start of the next frame and by every getter.
// the ONE host wait of a frame: spin on the sequence word the set-up kernel posts after the odometry iterations (the GPU goes straight on
// to integrate / raycast meanwhile)
static int wait_for_pose(kt_tracker* t)
{
    const auto w0 = std::chrono::steady_clock::now();
    volatile unsigned int* seq = &t->mirror->seq;
    long long spins = 0;
    while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != t->frame_seq) {
        if ((++spins & 0xfff) == 0) {
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            if (waited > 30.0) {
                const hipError_t e = hipStreamQuery(t->ctx->stream);
                kt_set_error("tracker: no pose from the device after 30 s (stream status: %s)", hipGetErrorString(e));
                return KT_ERR_STATE;
            }
            if (waited > 0.002) std::this_thread::yield();
        }
    }
    t->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    return KT_OK;
}

static int complete_frame(kt_tracker* t)
{
    if (!t->outstanding) return KT_OK;
    t->outstanding = false;
    kt_ctx* c = t->ctx;
    KT_TRY(wait_for_pose(t));
    if (t->out_ordinal + 1 > t->frames_observed) t->frames_observed = t->out_ordinal + 1;
    ev_collect(t);
    if (t->mirror->handoff_timeout) {
        // An inter-workgroup hand-off of this frame's odometry gave up (kt_track.hip): the frame has no pose, and the set-up kernel parked its
        // fusion (skip = 1).  The reference's icpStep is stream-ordered and cannot fail this way (reduce.cu:347-419), so neither may the frame:
        // the hand-off buffer is refilled (a sweep that gave up leaves it undefined: kt_refill_granules) and the odometry runs again, one
        // launch per iteration, from the frame's starting pose -- rmats_ / tvecs_.back() have not moved, the frame set and the previous
        // frame's predicted maps are untouched -- followed by the set-up kernel and the fusion exactly as process_frame enqueued them.
        // Only if THAT chain reports a time-out as well (no co-residency is involved: a fault, not contention) is the frame an error.
        KT_TRY(kt_refill_granules(c));
        ++t->odo_fallbacks;
        t->icp_demote = t->icp_demote_len;
        if (t->icp_demote_len < 65536) t->icp_demote_len *= 2;
        select_set(t, t->out_set);
        const bool icp = !(t->cfg.use_rgbd || t->cfg.use_rgbd_icp);
        if (icp) KT_TRY(icp_odometry(t, true));
        else KT_TRY(rgbd_odometry(t, t->out_set, t->out_last_set, true));
        if (++t->frame_seq == 0) t->frame_seq = 1;
        KT_TRY(launch_setup(t, 0, nullptr, nullptr));
        if (t->out_speculated) KT_TRY(enqueue_fusion(t, t->out_set, t->out_depth, t->out_rgb, t->plan_sel >= 0 ? &t->plans[t->plan_sel].plan : nullptr));
        KT_TRY(wait_for_pose(t));
        if (t->mirror->handoff_timeout) {
            (void)kt_refill_granules(c);   // before the caller's next frame (or its reset) launches another reduction
            kt_set_error("odometry: inter-workgroup hand-off timed out (also in the stepwise re-run of the frame)");
            return KT_ERR_STATE;
        }
    }
    float Rcurr[9], tcurr[3];
    memcpy(Rcurr, t->mirror->R, sizeof(Rcurr));
    memcpy(tcurr, t->mirror->t, sizeof(tcurr));
    if (t->pose_log_on) {   // test hook: the pose the set-up kernel of frame out_ordinal saw (before any shift of this frame)
        if (t->pose_log.size() < (size_t)(t->out_ordinal + 1) * 12) t->pose_log.resize((size_t)(t->out_ordinal + 1) * 12, 0.0f);
        memcpy(&t->pose_log[(size_t)t->out_ordinal * 12], Rcurr, sizeof(Rcurr));
        memcpy(&t->pose_log[(size_t)t->out_ordinal * 12 + 9], tcurr, sizeof(tcurr));
    }
    return finish_pose(t, Rcurr, tcurr, t->out_speculated);
}

extern "C" {

static int process_frame_impl(kt_tracker* t, const uint16_t* depth_raw, const uint8_t* colors, uint64_t timestamp);

int kt_tracker_process_frame(kt_tracker* t, const uint16_t* depth_raw, const uint8_t* colors, uint64_t timestamp)
{
    KT_ARG(t && depth_raw && colors);
    const auto c0 = std::chrono::steady_clock::now();
    const int r = process_frame_impl(t, depth_raw, colors, timestamp);
    t->host_call_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
    t->host_calls += 1;
    return r;
}

/* mean seconds per kt_tracker_process_frame call since the last reset of the statistics: {total in the call, of which waiting for
 * the previous frame's pose}; the difference is enqueue work */
int kt_tracker_host_times(kt_tracker* t, double out2[2], int reset)
{
    KT_ARG(t && out2);
    out2[0] = t->host_calls ? t->host_call_s / (double)t->host_calls : 0.0;
    out2[1] = t->host_calls ? t->host_wait_s / (double)t->host_calls : 0.0;
    if (reset) { t->host_call_s = t->host_wait_s = 0.0; t->host_calls = 0; }
    return KT_OK;
}

static int process_frame_impl(kt_tracker* t, const uint16_t* depth_raw, const uint8_t* colors, uint64_t timestamp)
{
    KT_TRY(complete_frame(t));
    kt_ctx* c = t->ctx;
    const bool gt = t->has_trajectory;
    if (gt && !t->trajectory.count((int)(uint32_t)timestamp)) {
        // GroundTruthOdometry::preRun :89-111, KintinuousTracker.cpp:460-463: a frame without a trajectory entry is not tracked at
        // all.  If it was read ahead, its set goes back to the pool once written.
        for (size_t i = 0; i < t->pending.size(); ++i)
            if (t->pending[i].depth == depth_raw && t->pending[i].rgb == colors) {
                KT_HIP(hipStreamWaitEvent(c->stream, t->sets[t->pending[i].set].ready, 0));
                KT_TRY(drop_plans_of_set(t, t->pending[i].set));
                t->pending.erase(t->pending.begin() + i);
                break;
            }
        return KT_OK;
    }
    t->current_ts = timestamp;
    const int cols = t->cfg.cols, rows = t->cfg.rows, N = t->N;
    const bool icp = !gt && !(t->cfg.use_rgbd || t->cfg.use_rgbd_icp);
    const int angle_color = !t->cfg.disable_color_angle;

    // [A] pyramid build, KintinuousTracker.cpp:465-479 (+ scaleDepth records): taken from the prefetch stream if this frame
    // was announced with kt_tracker_prefetch_frame, otherwise computed here
    int set = -1;
    bool read_ahead = false;
    for (size_t i = 0; i < t->pending.size(); ++i)
        if (t->pending[i].depth == depth_raw && t->pending[i].rgb == colors) {
            read_ahead = true;
            // read-aheads announced before this one were skipped by the caller: their sets return to the pool once written
            for (size_t j = 0; j < i; ++j) {
                KT_HIP(hipStreamWaitEvent(c->stream, t->sets[t->pending[j].set].ready, 0));
                KT_TRY(drop_plans_of_set(t, t->pending[j].set));
            }
            set = t->pending[i].set;
            t->pending.erase(t->pending.begin(), t->pending.begin() + i + 1);
            // a read-ahead that has already retired needs no device-side join (a wait packet is a 3-5 us bubble in front of the odometry)
            if (hipEventQuery(t->sets[set].ready) != hipSuccess) KT_HIP(hipStreamWaitEvent(c->stream, t->sets[set].ready, 0));
            break;
        }
    if (set < 0) {
        // not announced: build the set here (main-stream order protects the readers of whatever it overwrites)
        set = pick_free_set(t);
        if (set < 0) {  // both spare sets hold read-aheads the caller is not consuming: give up the older one
            set = t->pending.front().set;
            KT_HIP(hipStreamWaitEvent(c->stream, t->sets[set].ready, 0));
            KT_TRY(drop_plans_of_set(t, set));
            t->pending.erase(t->pending.begin());
        }
        t->last_assigned = set;
        KT_TRY(ev_begin(t, ST_PYRAMID));
        KT_TRY(build_frame_set(t, c, set, depth_raw, colors));
        KT_TRY(ev_end(t, ST_PYRAMID));
    }
    select_set(t, set);
    const long long ordinal = t->frames_started;
    t->sets[set].user = t->frames_started++;
    for (int k = 0; k < KT_NSLOTS; ++k)
        if (depth_raw == t->depth_stage[k]) t->slot_frame[k] = ordinal;
    const int last_set = t->prev_set;   // "last" of RGBDOdometry for this frame
    t->prev_set = set;
    t->out_set = set;
    t->out_last_set = last_set;
    t->carry_sel ^= 1;                  // the state the previous tracked frame left becomes "before this frame"

    if (t->global_time == 0) {  // [B] :481-557
        kt_mat33 Rcam, Rcam_inv;
        memcpy(Rcam.m, t->Rlast, sizeof(Rcam.m));
        kt_mat33_inverse(Rcam.m, Rcam_inv.m);
        const int empty[3] = {0, 0, 0};
        if (t->counting) KT_HIP(hipMemsetAsync(t->upd_dev, 0, 16 * sizeof(unsigned int), c->stream));
        KT_TRY(launch_setup(t, 2, nullptr, nullptr));   // colour-weight carry only
        KT_TRY(ev_begin(t, ST_INTEGRATE));
        tsdf23_hook_arm(t);
        KT_TRY(kt_integrate_tsdf_impl(c, depth_raw, cols, rows, &t->intr, t->volume_size, &Rcam_inv, t->tlast, t->tranc_dist, t->tsdf,
                                      t->depth_raw_scaled, empty, t->color, colors, t->nmaps_curr[0], angle_color, N,
                                      t->counting ? t->upd_dev : nullptr, t->rec_curr, nullptr, t->bricks, t->sets[set].dpmax));
        KT_TRY(ev_end(t, ST_INTEGRATE));
        for (int l = 0; l < KT_LEVELS; ++l)
            KT_TRY(kt_transform_maps(c, t->vmaps_curr[l], t->nmaps_curr[l], lvl_cols(t, l), lvl_rows(t, l), &Rcam, t->tlast, t->vmaps_g_prev[l],
                                     t->nmaps_g_prev[l]));
        ++t->global_time;
        t->ev_par ^= 1;
        t->gt_utime = timestamp;   // :527-528
        KT_HIP(hipEventRecord(t->odo_ev[ordinal % KT_NODO], c->stream));
        if (t->cfg.place_recognition) add_pr_sample(t);   // :546-549: the first frame is always sampled (pose index 0)
        push_pose(t, timestamp, t->Rlast, 1);
        if (t->counting) {
            unsigned int u = 0;
            KT_HIP(hipMemcpyAsync(&u, t->upd_dev, sizeof(u), hipMemcpyDeviceToHost, c->stream));
            KT_HIP(hipStreamSynchronize(c->stream));
            t->last_U = u;
            t->last_S = 0;
        }
        return KT_OK;
    }

    t->out_ts = timestamp;
    t->out_set = set;
    t->out_depth = depth_raw;
    t->out_rgb = colors;
    t->out_thresh = t->parked ? INT_MAX : t->cfg.voxel_shift;
    if (gt) {
        // the pose comes from the trajectory, on the host: nothing to defer, nothing to speculate on
        float Rcurr[9], tcurr[3];
        ev_collect(t);
        ground_truth_pose(t, timestamp, Rcurr, tcurr);
        t->gt_utime = timestamp;   // :574-575
        KT_HIP(hipEventRecord(t->odo_ev[ordinal % KT_NODO], c->stream));   // before this frame's fusion, after the previous one's
        KT_TRY(finish_pose(t, Rcurr, tcurr, false));
        t->ev_par ^= 1;
        if (t->counting) KT_TRY(read_counts(t));
        return KT_OK;
    }

    // the voxel kernel's task plan, if one was made for this frame while the previous one was tracked and the volume has not
    // shifted since (plan_ahead)
    t->plan_sel = -1;
    v_wrap_copy_update(t);
    // The plan of THIS frame, made right here from the last pose the host has seen plus ONE increment (round 6; rounds 4-5 planned the read-ahead
    // frame, two increments out, behind that frame's read-ahead).  It runs on the plan stream beside the previous frame's fusion kernels, has this
    // frame's whole odometry launch to finish, and predicts one step instead of two -- a constant-velocity error grows with the square of the
    // horizon: tighter margins, fewer misses (221 / 1 against 214 / 2 on the orbit), and a plan made AFTER a shift instead of one the shift invalidates.
    // (-d repositions the cube once the pose is known: nothing to plan for.)
    // Made by whichever call sees the previous pose first: kt_tracker_prefetch_frame of the NEXT frame, when the caller announces in front of
    // the hand-over (plan_frame_to_come: the plan is then on its stream before the read-ahead's four launches are enqueued, ~15 us of host time
    // earlier -- it has the previous frame's fusion kernels to run beside, and what is still queued when the odometry launch takes the machine
    // waits that launch out and delays the set-up kernel), or this call.
    if (read_ahead) {
        const kt_tracker::PlanSlot& pl = t->plans[ordinal % 3];
        // a plan made for this ordinal belongs to this frame only if it was made from this frame's set and buffers (the caller may hand over
        // another announced frame than the one that was next in line) for the storage wrap that still holds
        if (pl.ordinal == ordinal && pl.set == set && pl.depth == depth_raw && pl.rgb == colors && memcmp(pl.wrap, t->v_wrap_copy, sizeof(t->v_wrap_copy)) == 0)
            t->plan_sel = (int)(ordinal % 3);
        else
            KT_TRY(plan_frame(t, Pending{depth_raw, colors, set, nullptr, nullptr}, ordinal));
    }
    // [C] odometry :564-572 -- every Gauss-Newton iteration is enqueued; the pose stays on the device
    v_wrap_copy_update(t);
    if (++t->frame_seq == 0) t->frame_seq = 1;  // 0 is the mirror's initial value (before the odometry: its launch may carry the set-up, which posts it)
    t->setup_fused = false;
    KT_TRY(ev_begin(t, ST_ODOMETRY));
    if (icp) KT_TRY(icp_odometry(t));
    else KT_TRY(rgbd_odometry(t, set, last_set));
    // device-side frame set-up (which also posts the pose into the host's PoseMirror), then the fusion kernels -- enqueued right
    // here, speculatively, on the assumption that the volume does not shift
    v_wrap_copy_update(t);
    // the plan has had the 19 launches above to finish; a join is enqueued only if it has not (a wait packet is a bubble)
    if (t->plan_sel >= 0 && hipEventQuery(t->plans[t->plan_sel].done) != hipSuccess) KT_HIP(hipStreamWaitEvent(c->stream, t->plans[t->plan_sel].done, 0));
    // gated side streams: what they were given for the NEXT frame (its read-ahead, then its plan) ends before this frame's voxel kernel starts
    // -- one wait packet in front of the set-up kernel (3-5 us; the gate is on where the frame is a millisecond)
    if (t->side_gate && !t->pending.empty()) KT_HIP(hipStreamWaitEvent(c->stream, t->sets[t->pending.front().set].ready, 0));
    if (!t->setup_fused) KT_TRY(launch_setup(t, 0, nullptr, nullptr));
    // -d: the cube may be repositioned once the pose is known, which changes the shift decision -- nothing to speculate on
    t->out_speculated = !t->cfg.dynamic_cube;
    if (t->out_speculated) KT_TRY(enqueue_fusion(t, set, depth_raw, colors, t->plan_sel >= 0 ? &t->plans[t->plan_sel].plan : nullptr));
    t->outstanding = true;
    t->out_ordinal = ordinal;
    t->gt_utime = timestamp;
    if (!t->out_speculated) KT_TRY(complete_frame(t));   // observe the pose, reposition, shift if needed, enqueue the fusion
    t->ev_par ^= 1;
    if (t->counting) {  // the counters are read back per frame: finish it before returning
        KT_TRY(complete_frame(t));
        KT_TRY(read_counts(t));
    }
    return KT_OK;
}

static int prefetch_impl(kt_tracker* t, const uint16_t* depth_raw, const uint8_t* colors, const uint16_t* depth_host, const uint8_t* rgb_host);

int kt_tracker_prefetch_frame(kt_tracker* t, const uint16_t* depth_raw, const uint8_t* colors)
{
    KT_ARG(t && depth_raw && colors);
    return prefetch_impl(t, depth_raw, colors, nullptr, nullptr);
}

// stage a host frame into the next slot: pinned copy, then the upload on `stream`
static int stage_host_frame(kt_tracker* t, hipStream_t stream, const uint16_t* depth_host, const uint8_t* rgb_host, int* slot_out)
{
    const size_t P = (size_t)t->cfg.cols * t->cfg.rows;
    const int slot = t->next_slot;
    t->next_slot = (t->next_slot + 1) % KT_NSLOTS;
    KT_HIP(hipEventSynchronize(t->slot_uploaded[slot]));   // the pinned copy of the frame that used this slot 4 frames ago has left
    KT_TRY(wait_frame_consumed(t, stream, t->slot_frame[slot]));   // ... and its fusion has read the device copy
    t->slot_frame[slot] = -1;
    memcpy(t->depth_stage_host[slot], depth_host, P * sizeof(uint16_t));
    memcpy(t->rgb_stage_host[slot], rgb_host, P * 3);
    KT_HIP(hipMemcpyAsync(t->depth_stage[slot], t->depth_stage_host[slot], P * sizeof(uint16_t), hipMemcpyHostToDevice, stream));
    KT_HIP(hipMemcpyAsync(t->rgb_stage[slot], t->rgb_stage_host[slot], P * 3, hipMemcpyHostToDevice, stream));
    KT_HIP(hipEventRecord(t->slot_uploaded[slot], stream));
    *slot_out = slot;
    return KT_OK;
}

int kt_tracker_prefetch_frame_host(kt_tracker* t, const uint16_t* depth_host, const uint8_t* rgb_host)
{
    KT_ARG(t && depth_host && rgb_host);
    if (t->pending.size() >= 2) { kt_set_error("kt_tracker_prefetch_frame_host: two read-ahead frames are already outstanding"); return KT_ERR_STATE; }
    KT_TRY(complete_frame(t));
    int slot;
    KT_TRY(stage_host_frame(t, t->pre_stream, depth_host, rgb_host, &slot));   // the upload rides the read-ahead stream too
    return prefetch_impl(t, t->depth_stage[slot], t->rgb_stage[slot], depth_host, rgb_host);
}

static int prefetch_impl(kt_tracker* t, const uint16_t* depth_raw, const uint8_t* colors, const uint16_t* depth_host, const uint8_t* rgb_host)
{
    if (t->pending.size() >= 2) { kt_set_error("kt_tracker_prefetch_frame: two read-ahead frames are already outstanding"); return KT_ERR_STATE; }
    // Observe the pose of the frame in flight first (the next kt_tracker_process_frame call would wait for it anyway).  After that,
    // with F = frames handed over so far, everything enqueued before the end of odometry(F - 1) has retired: fusion(F - 2) and the
    // RGB-D "last" reads of odometry(F - 1).  So a set last consumed by frame <= F - 2 is free; the set of frame F - 1 (its fusion
    // may still run, and it is RGB-D "last" for frame F) and the sets of outstanding read-aheads are not.
    KT_TRY(complete_frame(t));
    // the frame that will be handed over next is known (announced earlier) and so is the pose in front of it: plan it NOW, before this frame's
    // read-ahead goes to its stream (process_frame_impl: "made by whichever call sees the previous pose first")
    if (!t->pending.empty() && t->plans[t->frames_started % 3].ordinal != t->frames_started) KT_TRY(plan_frame(t, t->pending.front(), t->frames_started));
    const int set = pick_free_set(t);   // exists: 3 sets, at most 1 other read-ahead outstanding here
    if (set < 0) { kt_set_error("kt_tracker_prefetch_frame: no free frame set"); return KT_ERR_STATE; }
    t->last_assigned = set;
    KT_TRY(wait_frame_consumed(t, t->pre_stream, t->sets[set].user));
    t->pre_ctx.device = t->ctx->device;
    if (t->side_gate && t->gate_armed) {
        // the frame just observed is in its voxel kernel now: hold the read-ahead (and, behind its `ready` event, the plan stream) until
        // that launch has drained -- the event enqueue_fusion recorded between the voxel kernel and the ray cast
        KT_HIP(hipStreamWaitEvent(t->pre_stream, t->gate_ev, 0));
    }
    KT_TRY(build_frame_set(t, &t->pre_ctx, set, depth_raw, colors));
    KT_HIP(hipEventRecord(t->sets[set].ready, t->pre_stream));
    t->pending.push_back(Pending{depth_raw, colors, set, depth_host, rgb_host});
    return KT_OK;
}

int kt_tracker_process_frame_host(kt_tracker* t, const uint16_t* depth_host, const uint8_t* rgb_host, uint64_t timestamp)
{
    // TrackerInterface::process upload, TrackerInterface.cpp:90-91 (pinned staging + async copies instead of blocking cudaMemcpy2D)
    KT_ARG(t && depth_host && rgb_host);
    for (const Pending& p : t->pending)   // announced with kt_tracker_prefetch_frame_host: already on the device
        if (p.depth_host == depth_host && p.rgb_host == rgb_host) return kt_tracker_process_frame(t, p.depth, p.rgb, timestamp);
    KT_TRY(complete_frame(t));
    int slot;
    KT_TRY(stage_host_frame(t, t->ctx->stream, depth_host, rgb_host, &slot));
    return kt_tracker_process_frame(t, t->depth_stage[slot], t->rgb_stage[slot], timestamp);
}

// KintinuousTracker::loadTrajectory (KintinuousTracker.cpp:216-260) minus the text parsing: pose7 = n x {x y z qx qy qz qw}.
// T.setIdentity(); T.pretranslate(t).rotate(q): linear = Quaternionf::toRotationMatrix() (no normalisation), translation = t.
int kt_tracker_load_trajectory(kt_tracker* t, int n, const uint64_t* utimes, const float* pose7)
{
    KT_ARG(t && n >= 0 && (n == 0 || (utimes && pose7)));
    KT_TRY(complete_frame(t));
    t->has_trajectory = true;
    for (int i = 0; i < n; ++i) {
        std::array<float, 12> T;
        kt_host_trajectory_pose(pose7 + (size_t)i * 7, T.data());
        t->trajectory[(int)(uint32_t)utimes[i]] = T;
    }
    t->gt_utime = 0;   // :259
    return KT_OK;
}

int kt_tracker_get_volume_basis(kt_tracker* t, float* basis)
{
    KT_ARG(t && basis);
    KT_TRY(complete_frame(t));
    memcpy(basis, t->volume_basis, sizeof(t->volume_basis));
    return KT_OK;
}

int kt_tracker_finalise(kt_tracker* t)
{
    // finalise :1003-1048
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    v_wrap_copy_update(t);
    const int lo[3] = {0, 0, 0}, hi[3] = {t->N, t->N, t->N};
    KT_TRY(fetch_slice(t, lo, hi, 7 /* CloudSlice::FINAL */));
    if (t->cfg.place_recognition) {   // :1035-1045: the final slice carries one more sample, taken at the last pose
        memcpy(t->pr_rot, t->Rlast, sizeof(t->pr_rot));
        memcpy(t->pr_trans, t->current_global_camera, sizeof(t->pr_trans));
        t->slices.back().pr_id = add_pr_sample(t);
        t->pr_samples.back().pose_index = (int)t->poses.size() - 1;   // no new pose follows: it belongs to the last one
    }
    return KT_OK;
}

/* (a negative count = the frame in flight could not be completed: kt_last_error() says why) */
int kt_tracker_num_pr_samples(kt_tracker* t) { return (t && complete_frame(t) == KT_OK) ? (int)t->pr_samples.size() : -1; }
int kt_tracker_pr_sample(kt_tracker* t, int i, uint64_t* utime, float* trans, float* rotation, int* pose_index)
{
    KT_ARG(t && utime && trans && rotation && pose_index);
    KT_TRY(complete_frame(t));
    KT_ARG(i >= 0 && i < (int)t->pr_samples.size());
    const PrSample& ps = t->pr_samples[i];
    *utime = ps.utime;
    memcpy(trans, ps.trans, sizeof(ps.trans));
    memcpy(rotation, ps.rot, sizeof(ps.rot));
    *pose_index = ps.pose_index;
    return KT_OK;
}
int kt_tracker_slice_pr_id(kt_tracker* t, int i, int* pr_id)
{
    KT_ARG(t && pr_id);
    KT_TRY(complete_frame(t));
    KT_ARG(i >= 0 && i < (int)t->slices.size());
    *pr_id = t->slices[i].pr_id;
    return KT_OK;
}

int kt_tracker_get_pose(kt_tracker* t, float* R, float* tv, float* gc)
{
    KT_ARG(t && R && tv && gc);
    KT_TRY(complete_frame(t));
    memcpy(R, t->Rlast, sizeof(t->Rlast));
    memcpy(tv, t->tlast, sizeof(t->tlast));
    memcpy(gc, t->current_global_camera, sizeof(t->current_global_camera));
    return KT_OK;
}
int kt_tracker_num_poses(kt_tracker* t) { return (t && complete_frame(t) == KT_OK) ? (int)t->poses.size() : -1; }
int kt_tracker_get_dense_pose(kt_tracker* t, int i, uint64_t* ts, float* pose16, int* is_loop)
{
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    KT_ARG(t && i >= 0 && i < (int)t->poses.size() && ts && pose16 && is_loop);
    *ts = t->poses[i].ts;
    memcpy(pose16, t->poses[i].pose, sizeof(t->poses[i].pose));
    *is_loop = t->poses[i].is_loop;
    return KT_OK;
}
int kt_tracker_get_voxel_wrap(kt_tracker* t, int* wrap)
{
    KT_ARG(t && wrap);
    KT_TRY(complete_frame(t));
    memcpy(wrap, t->voxel_wrap, sizeof(t->voxel_wrap));
    return KT_OK;
}
int kt_tracker_num_slices(kt_tracker* t) { return (t && complete_frame(t) == KT_OK) ? (int)t->slices.size() : -1; }
int kt_tracker_slice_info(kt_tracker* t, int i, size_t* n_points, int* dimension)
{
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    KT_ARG(t && i >= 0 && i < (int)t->slices.size() && n_points && dimension);
    KT_TRY(join_slice_jobs(t));
    *n_points = t->slices[i].pts.size();
    *dimension = t->slices[i].dim;
    return KT_OK;
}
int kt_tracker_set_parked(kt_tracker* t, int parked)
{
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    t->parked = parked != 0;
    return KT_OK;
}
int kt_tracker_slice_pose(kt_tracker* t, int i, float* R, float* cam, uint64_t* ts)
{
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    KT_ARG(t && i >= 0 && i < (int)t->slices.size());
    if (R) memcpy(R, t->slices[i].R, 9 * sizeof(float));
    if (cam) memcpy(cam, t->slices[i].cam, 3 * sizeof(float));
    if (ts) *ts = t->slices[i].ts;
    return KT_OK;
}
int kt_tracker_slice_points(kt_tracker* t, int i, kt_point_xyzrgb* out)
{
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    KT_ARG(t && i >= 0 && i < (int)t->slices.size() && out);
    KT_TRY(join_slice_jobs(t));
    if (!t->slices[i].pts.empty()) memcpy(out, t->slices[i].pts.data(), t->slices[i].pts.size() * sizeof(kt_point_xyzrgb));
    return KT_OK;
}
int16_t* kt_tracker_volume(kt_tracker* t) { return (t && complete_frame(t) == KT_OK) ? t->tsdf : nullptr; }
uint8_t* kt_tracker_color_volume(kt_tracker* t) { return (t && complete_frame(t) == KT_OK) ? t->color : nullptr; }
float* kt_tracker_vmap_g_prev(kt_tracker* t, int l) { return (t && l >= 0 && l < KT_LEVELS && complete_frame(t) == KT_OK) ? t->vmaps_g_prev[l] : nullptr; }
float* kt_tracker_nmap_g_prev(kt_tracker* t, int l) { return (t && l >= 0 && l < KT_LEVELS && complete_frame(t) == KT_OK) ? t->nmaps_g_prev[l] : nullptr; }
uint8_t* kt_tracker_vmap_curr_color(kt_tracker* t) { return (t && complete_frame(t) == KT_OK) ? t->vmap_curr_color : nullptr; }
float kt_tracker_trunc_dist(kt_tracker* t) { return t ? t->tranc_dist : 0.f; }

int kt_tracker_enable_profiling(kt_tracker* t, int on)
{
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    KT_HIP(hipStreamSynchronize(t->ctx->stream));
    ev_collect(t);
    t->profiling = on;
    memset(t->stage_ms_sum, 0, sizeof(t->stage_ms_sum));
    memset(t->stage_n, 0, sizeof(t->stage_n));
    return KT_OK;
}
int kt_tracker_stage_ms(kt_tracker* t, float* ms)
{
    KT_ARG(t && ms);
    KT_TRY(complete_frame(t));
    KT_HIP(hipStreamSynchronize(t->ctx->stream));
    ev_collect(t);
    for (int s = 0; s < ST_COUNT; ++s) ms[s] = t->stage_n[s] ? (float)(t->stage_ms_sum[s] / (double)t->stage_n[s]) : 0.f;
    return KT_OK;
}
int kt_tracker_stage_counts(kt_tracker* t, long long* n)
{
    KT_ARG(t && n);
    for (int s = 0; s < ST_COUNT; ++s) n[s] = t->stage_n[s];
    return KT_OK;
}
int kt_tracker_enable_counts(kt_tracker* t, int on)
{
    KT_ARG(t);
    t->counting = on;
    return KT_OK;
}
int kt_tracker_debug_state(kt_tracker* t, float* out29)
{
    KT_ARG(t && out29);
    KT_TRY(complete_frame(t));
    KT_HIP(hipStreamSynchronize(t->ctx->stream));
    KT_HIP(hipMemcpy(t->state_host, t->state_dev, sizeof(kt_track_state), hipMemcpyDeviceToHost));
    memcpy(out29, t->state_host->icp29, 29 * sizeof(float));
    return KT_OK;
}
/* The CloudSliceProcessor stage (weight cull, voxel grid at the voxel size, k-NN normals: kt_slice_process_device) behind every slab the
 * tracker extracts from now on, on a stream of its own; the processed points travel with the slice (kt_tracker_slice_processed). */
int kt_tracker_enable_slice_stage(kt_tracker* t, int on, int weight_cull, int k)
{
    KT_ARG(t && (!on || (k >= 1 && k <= 64)));
    KT_TRY(complete_frame(t));
    KT_TRY(join_slice_jobs(t));
    if (on && !t->slice_ws) {
        KT_TRY(kt_slice_ws_create(t->ctx, t->cloud_cap, nullptr, &t->slice_ws));
        for (int b = 0; b < 2; ++b) {
            KT_HIP(hipMalloc((void**)&t->cloud_dev[b], t->cloud_cap * sizeof(kt_point_xyzrgb)));
            KT_HIP(hipMalloc((void**)&t->cloud_count_dev[b], sizeof(unsigned int)));
            KT_HIP(hipHostMalloc((void**)&t->proc_host[b], t->cloud_cap * sizeof(kt_point_xyzrgbnormal), hipHostMallocDefault));
            KT_HIP(hipHostMalloc((void**)&t->proc_count_host[b], sizeof(unsigned int), hipHostMallocDefault));
            KT_HIP(hipEventCreateWithFlags(&t->extracted[b], KT_EV_DEVICE));
        }
        // the stage's kernels (the library sort needs scratch memory, which the runtime allocates at a kernel's first launch) run once
        // here, on an empty slab, so that the first shift does not pay for it
        KT_HIP(hipMemsetAsync(t->cloud_count_dev[0], 0, sizeof(unsigned int), (hipStream_t)kt_slice_ws_stream(t->slice_ws)));
        const float leaf = fmaxf(t->voxel_size[0], fmaxf(t->voxel_size[1], t->voxel_size[2]));
        KT_TRY(kt_slice_process_device(t->slice_ws, t->cloud_dev[0], t->cloud_count_dev[0], t->cloud_cap, weight_cull, leaf, k));
        KT_HIP(hipStreamSynchronize((hipStream_t)kt_slice_ws_stream(t->slice_ws)));
    }
    t->slice_stage = on != 0;
    t->slice_cull = weight_cull; t->slice_k = k;
    return KT_OK;
}
/* number of processed points of slice i, or -1 when the slice was extracted without the stage */
int kt_tracker_slice_processed_info(kt_tracker* t, int i, long long* n_points)
{
    KT_ARG(t && n_points);
    KT_TRY(complete_frame(t));
    KT_ARG(i >= 0 && i < (int)t->slices.size());
    KT_TRY(join_slice_jobs(t));
    *n_points = t->slices[i].has_processed ? (long long)t->slices[i].processed.size() : -1;
    return KT_OK;
}
int kt_tracker_slice_processed(kt_tracker* t, int i, kt_point_xyzrgbnormal* out)
{
    KT_ARG(t && out);
    KT_TRY(complete_frame(t));
    KT_ARG(i >= 0 && i < (int)t->slices.size());
    KT_TRY(join_slice_jobs(t));
    KT_ARG(t->slices[i].has_processed);
    if (!t->slices[i].processed.empty()) memcpy(out, t->slices[i].processed.data(), t->slices[i].processed.size() * sizeof(kt_point_xyzrgbnormal));
    return KT_OK;
}

int kt_tracker_odometry_fallbacks(kt_tracker* t, long long* out)
{
    KT_ARG(t && out);
    KT_TRY(complete_frame(t));
    *out = t->odo_fallbacks;
    return KT_OK;
}

int kt_tracker_plan_stats(kt_tracker* t, long long out2[2])
{
    KT_ARG(t && out2);
    KT_TRY(complete_frame(t));
    out2[0] = t->plan_hits; out2[1] = t->plan_misses;
    return KT_OK;
}
/* test hooks of the planned-ahead voxel pass (tests/test_gpu_tracker.py::test_plan_margins_*): */
int kt_tracker_debug_pose_log(kt_tracker* t, int enable, float* out12n, int max_frames, int* n_frames)
{
    KT_ARG(t);
    if (enable >= 0) t->pose_log_on = enable != 0;
    if (out12n || n_frames) {
        KT_TRY(complete_frame(t));
        const int n = (int)(t->pose_log.size() / 12);
        if (n_frames) *n_frames = n;
        if (out12n) memcpy(out12n, t->pose_log.data(), sizeof(float) * 12 * (size_t)(n < max_frames ? n : max_frames));
    }
    return KT_OK;
}
int kt_tracker_debug_plan_truth(kt_tracker* t, const float* poses12n, int n_frames, float fr, float ft, float theta_fixed, float tau_fixed, unsigned int seed)
{
    KT_ARG(t && n_frames >= 0 && (poses12n || n_frames == 0) && fr >= 0 && ft >= 0 && theta_fixed >= 0 && tau_fixed >= 0);
    t->plan_truth.assign(poses12n, poses12n + (size_t)n_frames * 12);
    t->plan_fr = fr; t->plan_ft = ft; t->plan_perturb = fr > 0.0f || ft > 0.0f;
    t->plan_theta_fixed = theta_fixed; t->plan_tau_fixed = tau_fixed;
    t->plan_rng = seed * 2654435761u + 0x9e3779b9u;
    return KT_OK;
}


int kt_tracker_debug_counts(kt_tracker* t, unsigned int* out4)
{
    KT_ARG(t && out4);
    KT_HIP(hipMemcpy(out4, t->upd_dev, 8 * sizeof(unsigned int), hipMemcpyDeviceToHost));
    { unsigned long long h[3] = {0, 0, 0}; KT_HIP(hipMemcpy(h, t->steps_dev + 1, sizeof(h), hipMemcpyDeviceToHost)); out4[7] = (unsigned int)h[0]; out4[5] = (unsigned int)h[1]; out4[6] = (unsigned int)h[2]; }
    return KT_OK;
}
int kt_tracker_last_counts(kt_tracker* t, unsigned long long* U, unsigned long long* S)
{
    KT_ARG(t && U && S);
    *U = t->last_U;
    *S = t->last_S;
    return KT_OK;
}

int kt_tracker_export_poses_device(kt_tracker* t, int k, float* dst_dev)
{
    KT_ARG(t);
    KT_TRY(complete_frame(t));
    KT_ARG(k > 0 && dst_dev && k <= (int)t->poses.size());
    std::vector<float> tmp((size_t)k * 16);
    for (int i = 0; i < k; ++i) memcpy(&tmp[(size_t)i * 16], t->poses[t->poses.size() - k + i].pose, 16 * sizeof(float));
    KT_HIP(hipMemcpyAsync(dst_dev, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice, t->ctx->stream));
    KT_HIP(hipStreamSynchronize(t->ctx->stream));
    return KT_OK;
}

}  // extern "C"

extern "C" int kt_tracker_debug_icp_levels(kt_tracker* t) { return t && t->last_icp_levels ? 1 : 0; }
extern "C" int kt_tracker_debug_side_gate(kt_tracker* t) { return t ? t->side_gate : 0; }
