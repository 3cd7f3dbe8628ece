// kt_setup.hpp -- the device-side frame set-up that runs behind a frame's last odometry iteration: it turns the device-resident Gauss-Newton
// result into what the fusion kernels need -- the final pose (RGB-D jump guard applied), its inverse, the z tables of tsdf23 (quirk A.17: a
// sequential float recurrence), the shift decision of KintinuousTracker.cpp:627-667, the colour-weight carry, the walk checkpoints of a planned
// voxel pass -- and posts the pose into the host's mirror.  Shared by its two homes: kt_frame_setup_kernel (kt_tracker.hip: a launch of its own,
// every odometry form) and, since round 6, the epilogue of kt_icp_level_kernel (kt_track.hip: the ICP chain's single launch ends with it -- one
// kernel boundary and the 256 otherwise idle workgroups' worth of time less per frame).
#pragma once

#include "kt_internal.hpp"
#include "kt_track.hpp"

#include <limits.h>

// the host's window on the frame in flight: written straight into pinned, device-mapped host memory (payload, system-scope fence, then seq),
// polled by complete_frame() -- no copy, no event, no driver call on the critical path
struct kt_pose_mirror {
    float R[9], t[3];
    int skip, handoff_timeout;
    unsigned int seq;
};

__host__ __device__ static inline int voxel_trans(float translation, float voxel, int thresh)
{
    // KintinuousTracker.cpp:640-667
    const int f = (int)floorf(translation / voxel);
    if (f < 0) return (-thresh > f) ? -thresh : f;
    return thresh < f ? thresh : f;
}

// mode 0: the pose comes from the odometry; mode 1 = pose supplied by the host (the redo of a parked frame); mode 2 = carry only (first frame)
struct kt_setup_args {
    kt_track_state* st; kt_frame_params* fp;
    kt_pose_mirror* mirror; unsigned int seq;
    float* vgz; float* zs; int N; float cell_z;
    int mode, rgbd_guard;
    float R[9], t[3];
    float basis[3], voxel[3]; int thresh;
    kt_pixrec* rec; const float* carry_cur; float* carry_next; int npix;
    // planned frames (kt_volume.hip "planning ahead"): the prediction and margins the plan was made with -- the pose is checked against
    // them -- and what the checkpoint workgroups need: the plan's wave-column ranges, where the checkpoints go, the walk's constants
    int carry_groups;
    const unsigned int* plan_wrange; float2* plan_walk0;
    float plan_R[9], plan_t[3], plan_theta, plan_tau;
    int wx, wy, wcx, wcy, XG, YG;        // storage wrap (x, y), wave-column shape and grid
    float cell_x, cell_y, fx, fy;
    int walk_groups;                     // checkpoint workgroups (0 without a plan); virtual blocks = 1 + carry_groups + walk_groups
    int fused;                           // kt_icp_level_kernel: run the set-up in the launch's epilogue (the tracker then enqueues no kt_frame_setup_kernel)
};

// the pose the frame is fused with: the odometry's result, or the previous pose when the RGB-D jump guard discards the increment
// (RGBDOdometry.cpp:383-387).  A pure function of the tracking state: every workgroup that needs it computes the same bits.
__device__ __forceinline__ void kt_setup_final_pose(const kt_setup_args& a, float R[9], float tv[3])
{
    for (int k = 0; k < 9; ++k) R[k] = a.st->Rcurr[k];
    for (int k = 0; k < 3; ++k) tv[k] = a.st->tcurr[k];
    if (a.rgbd_guard) {
        const float d0 = tv[0] - a.st->tprev[0], d1 = tv[1] - a.st->tprev[1], d2 = tv[2] - a.st->tprev[2];
        if ((double)__builtin_sqrtf(d0 * d0 + d1 * d1 + d2 * d2) > 0.3) {
            for (int k = 0; k < 9; ++k) R[k] = a.st->Rprev[k];
            for (int k = 0; k < 3; ++k) tv[k] = a.st->tprev[k];
        }
    }
}

// Virtual block vb >= 1 of the set-up (256 threads, ltid = 0..255), given the frame's final pose (mode 0; unused by the carry blocks):
// vb <= carry_groups: the colour-weight carry of KT_REC_STALE_NZ pixels, 1024 pixels each -- a pixel without a valid normal takes the weight the
// carry holds (and passes it on), every other pixel deposits its own; a pure function of (rec flags, rec.wrkc of valid pixels, carry_cur), so
// running it twice for a frame changes nothing.  vb > carry_groups: checkpoint blocks of a planned frame, one wave per wave-column of the plan:
// the walk of v_x, v_y from z = 0 to the wave-column's first z for its 64 columns (the checkpoints are DEFINED by the frame's own pose, quirk A.17).
__device__ __forceinline__ void kt_setup_side_block(const kt_setup_args& a, int vb, int ltid, const float (&R)[9], const float (&tv)[3])
{
    if (vb > a.carry_groups) {
        const int w = (vb - 1 - a.carry_groups) * 4 + (ltid >> 6), lane = ltid & 63;
        if (w >= a.XG * a.YG) return;
        const unsigned int r = a.plan_wrange[w];
        const int zc = (int)(r & 0xffffu);
        if (zc >= (int)(r >> 16)) return;   // no task in this wave-column
        const int sx = (w % a.XG) * a.wcx + lane % a.wcx, sy = (w / a.XG) * a.wcy + lane / a.wcx;
        if (sx >= a.N || sy >= a.N) return;
        float Rinv[9];
        kt_mat33_inverse(R, Rinv);
        a.plan_walk0[(size_t)sy * a.N + sx] = kt_tsdf_walk_checkpoint(Rinv, tv[0], tv[1], tv[2], a.cell_x, a.cell_y, a.cell_z, a.fx, a.fy, sx, sy, a.wx, a.wy, a.N, zc);
        return;
    }
    const int base = (vb - 1) * 1024 + ltid;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = base + k * 256;
        if (p < a.npix) {
            const bool stale = (a.rec[p].rgbf & KT_REC_STALE_NZ) != 0;
            const float w = stale ? a.carry_cur[p] : a.rec[p].wrkc;
            a.carry_next[p] = w;
            if (stale) a.rec[p].wrkc = w;
        }
    }
}

// Block 0, the set-up proper, on two waves (role = wave, lane = 0..63), each from its own copy of the (cheap, deterministic) pose arithmetic:
// role 0 writes what the device-side consumers read and walks the z tables; role 1 tells the host -- its system-scope fence costs 2-3 us and would
// otherwise sit in front of the walk.  Rin / tin: the frame's final pose (mode 0: from the odometry; mode 1: the host's); timeout: the odometry's
// hand-off gave up (no pose: nothing may be fused with it, complete_frame re-runs the odometry).
__device__ __forceinline__ void kt_setup_block0(const kt_setup_args& a, const float (&Rin)[9], const float (&tin)[3], int timeout, int role, int lane)
{
    if (role == 1 && a.mode != 0) return;
    float R[9], tv[3];
    for (int k = 0; k < 9; ++k) R[k] = Rin[k];
    for (int k = 0; k < 3; ++k) tv[k] = tin[k];
    int skip = 0;
    if (a.mode == 0) {
        for (int k = 0; k < 3; ++k) {
            const int vt = voxel_trans(tv[k] - a.basis[k], a.voxel[k], a.thresh);
            if (vt >= a.thresh || vt <= -a.thresh) skip = 1;
        }
        if (timeout) skip = 1;   // no pose: nothing may be fused with it (complete_frame re-runs the frame's odometry)
        if (a.plan_wrange && !skip) {
            // The plan is conservative for every pose within plan_theta (rotation) and plan_tau (translation) of the prediction:
            // |R - R^|_F = 2 sqrt(2) sin(angle / 2) <= sqrt(2) angle.  Outside: skip = 2, the host fuses the frame through the in-stream
            // pre-pass instead (the enqueued voxel kernel and ray cast do nothing).
            float dr = 0.0f, dt = 0.0f;
            for (int k = 0; k < 9; ++k) dr += (R[k] - a.plan_R[k]) * (R[k] - a.plan_R[k]);
            for (int k = 0; k < 3; ++k) dt += (tv[k] - a.plan_t[k]) * (tv[k] - a.plan_t[k]);
            if (!(__builtin_sqrtf(dr) <= 1.40f * a.plan_theta && __builtin_sqrtf(dt) <= 0.99f * a.plan_tau)) skip = 2;
        }
    }
    if (role == 1) {
        if (lane == 0) {
            // what the host needs: the final pose and whether the fusion kernels run -- straight into its memory
            for (int k = 0; k < 9; ++k) a.mirror->R[k] = R[k];
            for (int k = 0; k < 3; ++k) a.mirror->t[k] = tv[k];
            a.mirror->skip = skip;
            a.mirror->handoff_timeout = timeout;
            __threadfence_system();
            __hip_atomic_store(&a.mirror->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    float Rinv[9];
    kt_mat33_inverse(R, Rinv);
    if (lane == 2) {
        for (int k = 0; k < 9; ++k) { a.fp->R[k] = R[k]; a.fp->Rinv[k] = Rinv[k]; }
        for (int k = 0; k < 3; ++k) a.fp->t[k] = tv[k];
        a.fp->skip = skip;
        // (the tracking state's pose is left as the odometry wrote it: the next frame starts from the host's copy of the final pose)
        if (a.mode == 0) a.st->fusion_skipped = skip;
    }
    // lane 0 walks v_g_z, lane 1 walks z_scaled: the same dependent float adds as tsdf23's z loop (tsdf_volume.cu:560-640),
    // 16 at a time in registers so the chain runs at add latency
    if (lane < 2 && !skip) {
        float acc = lane == 0 ? __builtin_fmaf(0 + 0.5f, a.cell_z, -tv[2]) : 0.0f;
        float* tab = lane == 0 ? a.vgz : a.zs;
        int z = 0;
        for (; z + 16 <= a.N; z += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { v[u] = acc; acc += a.cell_z; }
#pragma unroll
            for (int u = 0; u < 16; u += 4) *(float4*)&tab[z + u] = make_float4(v[u], v[u + 1], v[u + 2], v[u + 3]);
        }
        for (; z < a.N; ++z) { tab[z] = acc; acc += a.cell_z; }
    }
}
