/* kt_debug.h -- test and analysis hooks of libkt_hip.so (the measurement kernels -- PMC calibration streams, instruction issue rates, the
 * exhaustive division check -- live in libkt_debug.so: csrc/kt_measure.h).  NOT part of the drop-in boundary (include/kt_abi.h): nothing
 * here replaces a reference interface; the entry points exist for tests/, scripts/ and bench.py's diagnostics (kernel name, counters)
 * and may change without notice.  They are exported by the same library because they must run the product's own device code
 * (the voxel kernel's reciprocal chain, the Gauss-Newton tail, the granule hand-off ...) rather than a copy of it. */
#ifndef KT_DEBUG_H
#define KT_DEBUG_H

#include "../../include/kt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostics of the last counted integrate: {U, wave batches of 4 z-steps, active wave-chunks, sum and max of the active
 * waves' durations, sum of their issue and consume phases (10 ns ticks), 0} */
int kt_tracker_debug_counts(kt_tracker* t, unsigned int out8_host[8]);
/* diagnostics: the 29 ICP sums stashed by the last joint RGB-D + ICP iteration (or timing probes in instrumented builds) */
int kt_tracker_debug_state(kt_tracker* t, float out29_host[29]);
/* test hooks of the voxel pass planned ahead of its frame (csrc/kt_tracker.hip plan_ahead; tests/test_gpu_tracker.py):
 * kt_tracker_debug_pose_log: enable >= 0 switches the log of the poses the frames' set-up kernels saw (12 floats per frame: R row-major,
 * t; before that frame's own shift) on or off; out12n / n_frames, when given, receive it.
 * kt_tracker_debug_plan_truth: poses12n (from the log of an identical earlier run) replace the motion extrapolation as the prediction
 * of every frame they cover; the prediction is then offset by a rotation of fr * theta about a random axis and by ft * tau along a random
 * direction (theta, tau = the plan's margins; fixed to theta_fixed / tau_fixed when those are > 0).  A plan whose frame lands at
 * fr, ft < 0.99 of its margins must be accepted and give the same volume as no plan; beyond 1 it must be rejected. */
int kt_tracker_debug_pose_log(kt_tracker* trk, int enable, float* out12n, int max_frames, int* n_frames);
int kt_tracker_debug_plan_truth(kt_tracker* trk, const float* poses12n, int n_frames, float fr, float ft, float theta_fixed, float tau_fixed, unsigned int seed);
/* test / A-B hook: selects the voxel kernel of kt_integrate_tsdf and the tracker for N < 1024: 1 = kt_tsdf23_lean_kernel (round 4), 0 = the
 * round-3 kernel, -1 = back to the default (KT_TSDF_LEAN in the environment, else the build's).  Both store the same bits. */
int kt_debug_tsdf_lean(int on);
/* the arithmetic contract of the lean voxel kernel: 0 = bit-exact (default), 1 = "survey-8c" (kt_tsdf23_tol_kernel: SURVEY.md 8(c)'s parity
 * policy -- tsdf shorts within 1, colour bytes within 1 at rounding ties, voxel / pixel indices exact; weights exact except where the update
 * predicate `sdf >= -trunc` is decided by the last bit of v_sqrt_f32: a voxel exactly at the -trunc edge may be updated by one kernel and not
 * by the other -- 0 to 2 voxels of 135 M in tests/test_gpu_tol.py, a count, not a guarantee), 2 = "speed of light" (kt_tsdf23_sol_kernel: a
 * MEASUREMENT variant with the per-voxel arithmetic the reference's --prec-div=false --prec-sqrt=false build would execute; its results are not
 * the reference's and nothing but bench.py's roofline*.speed_of_light times it), -1 = environment / default */
int kt_debug_tsdf_contract(int tol);
const char* kt_debug_tsdf_kernel(void);   /* name of the voxel kernel the next N < 1024 launch uses (bench.py reports it) */
/* analysis hook (builds with -DKT_ICP_TIMING only; otherwise KT_ERR_STATE): per workgroup of the last reduction launch, 100 MHz stamps:
 * [0, 256) pixel loop entered, [256, 512) loop done, [512, 768) granules published */
int kt_debug_icp_wg_times(kt_ctx* ctx, unsigned long long* out768_host);
/* analysis hook (builds with -DKT_TSDF_TIMELINE only; otherwise a negative status): per wave of the last voxel-kernel launch
 * {HW_ID, 100 MHz stamps: entry, tables ready, then per task: set up, after every batch; exit}; returns the words per wave */
int kt_debug_tsdf_timeline(kt_ctx* ctx, unsigned long long* out_host, int max_words);
/* test hook: the Gauss-Newton tail of the reduction kernels (6x6 pivoted LDL^T in double, cv::Rodrigues, the pose composition;
 * ICPOdometry.cpp:127-178) in its two device forms -- one thread, and spread over the lanes of a wave (the one the kernels run) -- on n
 * caller-supplied systems.  cases_host: n records {float packed[32] (reduce.cu:401-418 order), float packed2[32], double resultRt[16],
 * float Rprev[9], float tprev[3], int joint, int pad[3]}; serial_out_host / wave_out_host: n device state records each; layout_out =
 * {sizeof(state record), offsets of resultRt (16 doubles), Rcurr (9 floats), tcurr (3 floats), sizeof(case record)} (n = 0: layout only). */
int kt_debug_solve_check(kt_ctx* ctx, int n, const void* cases_host, void* serial_out_host, void* wave_out_host, int layout_out[5]);
/* test hook: out[v + 32768] = the device's unpack_tsdf(v) for every short v (device.hpp:77-83 restated without a division) */
int kt_debug_unpack_table(kt_ctx* ctx, float* out_host65536);
/* test hook: number of floats d, 2^-20 <= |d| <= 2^20, for which the voxel kernel's unwrapped reciprocal chain differs from 1.0f / d */
int kt_debug_rcp_check(kt_ctx* ctx, unsigned int* mismatches_host);
/* test hook (csrc/kt_track.hip): after `skip` more ICP reduction launches on the context, `count` launches lose a publishing workgroup, so
 * their hand-off sweep gives up after spin_limit looks (0: unchanged) and the caller is told KT_ERR_STATE; dirty_out (optional) = number
 * of reduction granules that are not the sentinel once the context's stream has drained (0 = the buffer is clean for the next launch) */
int kt_debug_handoff_fault(kt_ctx* ctx, int skip, int count, unsigned int spin_limit, unsigned int* dirty_out);
/* test / tuning hook: the time bound of the hand-off waits (sweep; the level kernel's pose wait takes twice as long) in ticks of the 100 MHz
 * clock; 0 = the default, 5 000 000 = 50 ms */
int kt_debug_wait_limit(kt_ctx* ctx, unsigned int ticks_100mhz);
/* A/B hook: trackers created from now on run their ICP-only odometry as one launch per pyramid level (csrc/kt_track.hip: kt_icp_level_kernel) -- 1 --
 * or as one launch per iteration -- 0; -1 = KT_ICP_LEVELS in the environment, else the build's default.  Both give the same bits. */
int kt_debug_icp_levels(int on);
/* A/B hook: trackers created in -ri mode run their joint RGB-D + ICP odometry as one launch per pyramid level (kt_joint_level_kernel) -- 1 -- or as two
 * launches per iteration -- 0; -1 = KT_RI_LEVELS in the environment, else off (the level form is bit-equal and measured slower: kt_tracker.hip) */
int kt_debug_ri_levels(int on);
int kt_tracker_debug_icp_levels(kt_tracker* trk);   /* 1: the tracker's last frame ran its ICP chain in the level form (only while it is the process's only live tracker) */
int kt_tracker_debug_side_gate(kt_tracker* trk);    /* 1: the tracker's read-ahead waits for the ray cast of the frame in flight (KT_SIDE_GATE; csrc/kt_tracker.hip) */
/* host arithmetic behind the ICP row's threshold tests (csrc/kt_track.hip: kt_icp_set_thresholds): the largest float X with
 * sqrtf(X) <= T (strict = 0) or sqrtf(X) < T (strict = 1), -1 when there is none */
float kt_debug_sq_threshold(float T, int strict);

#ifdef __cplusplus
}
#endif

#endif
