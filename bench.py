#!/usr/bin/env python
"""bench.py -- RGB-D frames/s of the per-frame tracking + fusion hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 one rank per GPU under torch.distributed.run -- started by
the caller (RANK / WORLD_SIZE in the environment; WORLD_SIZE must equal --gpus) or, when the command is run bare, by bench.py itself,
which re-executes under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (resolve_launch).  A "step" is ONE frame through KintinuousTracker::processFrame (pyramid build, 19 ICP Gauss-Newton
iterations solved on the device, shift check, TSDF integrate, raycast, predicted-map pyramid) with the frame already
resident in HBM.  Workload at every N: BASELINE.json configs[1] -- 640x480 synthetic orbit, ICP-only tracking, 512^3
TSDF -- one independent stream per GPU (seed 1234 + rank), poses gathered once with an RCCL all_gather (weak scaling).
Prints ONE JSON line on rank 0.  `roofline` = the voxel kernel (tsdf::integrate) of the timed frames, HIP events on its launch stream, against 8 TB/s; `roofline.contract`
names the arithmetic contract those frames ran under (bit-exact unless --contract says otherwise), `roofline.contract_ab` the same launch alone under both contracts;
`roofline_stress` = BASELINE configs[4] with `frac_alone`, `frac_pipelined` and `survey8c` labelled; `cpu_baseline` = the oracle on this host's cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """librccl prints a version banner through C stdio on stdout (at init or at its first collective); the driver reads ONE JSON line
    from stdout.  While RCCL code runs, file descriptor 1 points at stderr, and C stdio is flushed before it is restored."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs (= ranks = independent streams) of the job; default: WORLD_SIZE, else 1")
    ap.add_argument("--launch-check", action="store_true",
                    help="resolve the launch (re-exec under torch.distributed.run if needed), meet the other ranks in the key-value store, print "
                         "{n_gpus, ranks} on rank 0 and exit: no GPU is touched (tests/test_host_logic.py)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="orbit512", choices=["orbit512", "orbit256", "crabwalk512", "farwall768"])
    ap.add_argument("--unique-frames", type=int, default=120, help="frames rendered; the trajectory is played ping-pong beyond that")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-readahead", action="store_true", help="process frames strictly one at a time (no kt_tracker_prefetch_frame)")
    ap.add_argument("--host-frames", action="store_true",
                    help="frames start in host memory (pinned staging copy + PCIe upload inside the timed region): the PCIe-inclusive rate, not the headline")
    ap.add_argument("--cpu-frames", type=int, default=31, help="frames of the CPU baseline sample (about 3 s wall = 100 core-seconds on 32 threads)")
    ap.add_argument("--contract", default="bit-exact", choices=["bit-exact", "survey-8c"],
                    help="arithmetic contract of the voxel kernel in the timed frames: bit-exact (kt_tsdf23_lean_kernel; the default and the one every parity "
                         "test holds) or survey-8c (kt_tsdf23_tol_kernel: SURVEY.md 8(c)'s parity policy, counted by tests/test_gpu_tol.py)")
    ap.add_argument("--no-contract-ab", action="store_true", help="skip the untimed A/B of the voxel kernel's two contracts (profiling runs: keeps their launches out of the kernel statistics)")
    ap.add_argument("--no-stress", action="store_true", help="skip the roofline_stress block (BASELINE.json configs[4]: 1280x960 @ 768^3, 3 frames)")
    return ap.parse_args()


WORKLOADS = {
    # name: (synth config, image scale, N, tracker kwargs)
    "orbit512": ("orbit", 1, 512, {}),
    "orbit256": ("orbit", 1, 256, {}),
    "crabwalk512": ("crabwalk", 1, 512, dict(volume_size=7.0, use_rgbd_icp=1)),
    "farwall768": ("farwall", 2, 768, dict(static_mode=1)),
}


# which BASELINE.json config a workload is (every line used to say configs[1])
BASELINE_CONFIG = {"orbit512": "BASELINE.json configs[1]", "crabwalk512": "BASELINE.json configs[2]", "farwall768": "BASELINE.json configs[4]",
                   "orbit256": "configs[0]-sized volume; not a BASELINE bench line"}


from kintinuous_amd.multistream import check_gather, make_comm, make_exchange, pingpong, stream_seed, timed_region  # noqa: E402


SLICE_NAMES = {0: "X+", 1: "X-", 2: "Y+", 3: "Y-", 4: "Z+", 5: "Z-", 7: "FINAL"}   # CloudSlice::Dimension


def resolve_launch(gpus, environ):
    """How `--gpus` and the launcher's environment combine (`--gpus` was parsed and never read, so a bare `bench.py --gpus 8`
    ran ONE stream).  Returns ("run", rank, local_rank, world) when this process is a rank of the job, or ("exec", world) when it must
    replace itself by torch.distributed.run with `world` ranks.  A launcher's WORLD_SIZE that disagrees with --gpus is an error: the JSON
    line's n_gpus must equal --gpus in every launch form."""
    launched = "WORLD_SIZE" in environ and "RANK" in environ
    if launched:
        world = int(environ["WORLD_SIZE"])
        if gpus is not None and gpus != world:
            raise SystemExit(f"bench: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
        return ("run", int(environ["RANK"]), int(environ.get("LOCAL_RANK", environ["RANK"])), world)
    if gpus is None or gpus == 1:
        return ("run", 0, 0, 1)
    if gpus < 1:
        raise SystemExit(f"bench: --gpus {gpus}")
    return ("exec", gpus)


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def main():
    args = parse()
    how = resolve_launch(args.gpus, os.environ)
    if how[0] == "exec":   # a bare `python bench.py --gpus N`: become the N-rank job the driver's torchrun line would have started
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={how[1]}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, env)
    _, rank, local_rank, world = how
    if args.launch_check:
        ex = make_exchange(rank, world)
        ex.share("built", b"1")
        top = ex.max("rank", float(rank))
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "highest_rank_seen": int(top), "gpus_flag": args.gpus}))
        return
    # One communicator per job: the C-ABI's (kt_comm over RCCL) does the pose gather AND the barriers around the timed region; rank 0's
    # communicator id and the max-over-ranks time travel through a key-value store (no torch process group, no second communicator).
    exchange = make_exchange(rank, world)

    from kintinuous_amd import abi, build as kbuild, synth
    if rank == 0 and kbuild.needs_build():
        kbuild.build()
    exchange.share("built", b"1")   # the other ranks load the library only once rank 0 has (re)built it

    cfg_name, scale, N, kw = WORKLOADS[args.workload]
    N = int(os.environ.get("KT_BENCH_N", N))  # experiments only
    cam = synth.Camera.scaled(scale)
    total_frames = args.steps + args.warmup
    nuniq = max(2, min(args.unique_frames, total_frames))
    seed = stream_seed(rank)
    _, frames, traj, kw2 = synth.sequence(cfg_name, nuniq, cam, seed)
    d = dict(volume_size=6.0, voxel_shift=14, overlap=2, static_mode=0, use_rgbd=0, use_rgbd_icp=0, fast_odometry=0, disable_color_angle=0)
    d.update(kw2)
    d.update(kw)

    ctx = abi.Ctx(local_rank)
    if args.contract == "survey-8c":
        abi._chk(abi.lib().kt_debug_tsdf_contract(1))
    cfg = abi.TrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, d["volume_size"], d["voxel_shift"], d["overlap"], d["static_mode"],
                            d["use_rgbd"], d["use_rgbd_icp"], d["fast_odometry"], d["disable_color_angle"], 0)
    trk = abi.Tracker(ctx, cfg)
    dev_frames = [(ctx.upload(dep), ctx.upload(rgb)) for (dep, rgb) in frames]  # inputs resident in HBM before the timed region

    # log playback with one frame of read-ahead: frame i + 1 is announced before frame i is handed over, so its pose-independent
    # stages (bilateral, pyramids, scaleDepth) run on the tracker's second stream -- throttled to 2 workgroups per CU -- under
    # frame i's latency-bound odometry chain.  Every stage of every frame still executes inside the timed region (the first timed
    # frame's read-ahead is issued in the warm-up, the last timed frame issues one for a frame after the region: the counts balance).
    readahead = not args.no_readahead

    host_frames = [(np.ascontiguousarray(dep, np.uint16), np.ascontiguousarray(rgb, np.uint8)) for (dep, rgb) in frames]

    def step(i, announce_next=True):
        if args.host_frames:
            if readahead and announce_next:
                trk.prefetch_frame_host(*host_frames[pingpong(i + 1, nuniq)])
            trk.process_frame_host(*host_frames[pingpong(i, nuniq)], 33333 * i)
            return
        if readahead and announce_next:   # frame i + 1's read-ahead gets the whole of frame i's odometry chain to hide under
            nd, nr = dev_frames[pingpong(i + 1, nuniq)]
            trk.prefetch_frame(nd, nr)
        dd, dr = dev_frames[pingpong(i, nuniq)]
        trk.process_frame(dd, dr, 33333 * i)

    # the one collective of the path: an RCCL all-gather of the ranks' dense poses through the C-ABI (kt_comm_*), also with a single
    # rank (a one-rank communicator: the same call path, so the default run exercises it); KT_BENCH_NO_COMM=1 switches it off
    comm = None
    if not os.environ.get("KT_BENCH_NO_COMM"):
        try:
            with stdout_to_stderr():
                comm = make_comm(exchange, ctx, rank, world)
        except Exception as e:   # no librccl on this host: the single-GPU measurement does not depend on it
            if world > 1:
                raise
            sys.stderr.write(f"bench: pose gather disabled ({e})\n")
    # HIP events around the tsdf23 kernel only, on the stream it is launched on: one frame in 2 when the region is very short (the driver's
    # 20 frames: 10 samples), one in 4 up to 50 frames (an event pair is two marker packets = ~10 us of bubbles in a 340 us frame: on every
    # frame that is 3 % of the rate being measured), one in 8 otherwise.
    def prepare():   # between the warm-up and the first barrier
        trk.enable_profiling(6 if args.steps <= 30 else (5 if args.steps <= 50 else 1))
        trk.host_times(reset=True)

    def gather():   # the single RCCL gather of per-stream poses, inside the timed region
        with stdout_to_stderr():
            return comm.gather_poses(trk, min(args.steps, trk.num_poses()))

    if world > 1 and comm is None:
        raise RuntimeError("bench: a multi-GPU run needs the pose communicator (librccl)")
    with stdout_to_stderr() if world > 1 else contextlib.nullcontext():   # (RCCL prints its banner at the first collective)
        region = timed_region(comm, exchange, world, ctx.sync, step, args.steps, args.warmup, gather if comm is not None else None, prepare)
    marks, allp = region["marks"], region["gathered"]
    pose_bytes = int(allp.size * 4) if allp is not None else 0
    k = allp.shape[1] if allp is not None else 0
    elapsed = region["elapsed"]   # the slowest rank's

    timed_poses = [trk.dense_pose(trk.num_poses() - k + i)[1] for i in range(k)] if comm is not None else []
    first_poses = [trk.dense_pose(i)[1] for i in range(min(trk.num_poses(), args.cpu_frames))]   # frames 0.. of the sequence (warm-up first)
    if os.environ.get("KT_BENCH_DEBUG"):
        sys.stderr.write(f"bench: num_poses {trk.num_poses()} first_poses {len(first_poses)} cpu_frames {args.cpu_frames}\n")
    # per-frame period seen by the caller (shift frames show up as the tail: slab extraction + download + clears on the host path)
    periods = np.diff(np.array(marks)) * 1e3
    slices_by_dim = {}
    for si in range(trk.num_slices()):
        _, sdim = trk.slice_info(si)
        slices_by_dim[SLICE_NAMES.get(sdim, str(sdim))] = slices_by_dim.get(SLICE_NAMES.get(sdim, str(sdim)), 0) + 1
    host_call_s, host_wait_s = trk.host_times()
    fallbacks = trk.odometry_fallbacks()
    stage = trk.stage_ms()
    tsdf23_ms, tsdf23_n = stage["tsdf23"]
    # tracking must still be healthy at the end of the timed region (not a degenerate run)
    R, t, _ = trk.pose()
    Rg, cg = traj[pingpong(args.warmup + args.steps - 1, nuniq)]
    basis = np.array([d["volume_size"] / 2] * 3)
    if d["static_mode"]:
        basis[2] = np.float32(d["volume_size"] * 0.5) - np.float32(d["volume_size"] * 0.5 + 0.45)
    w = trk.voxel_wrap().astype(np.float64) * (d["volume_size"] / N)
    pose_err = float(np.abs((t + w) - (cg + basis)).max())

    # ---- untimed: per-stage breakdown (events around every stage), then the U / S counters of a few frames -----------
    base = args.warmup + args.steps
    stage_pipe = None
    if readahead:
        # the stages of the main stream as the timed frames run them: read-ahead on, the voxel pass from the task plan made ahead of the
        # frame (events around every stage: each one costs a bubble, which is why this is not done inside the timed region)
        step(base, announce_next=True)
        trk.enable_profiling(2)
        for i in range(base + 1, base + 9):
            step(i, announce_next=True)
        stage_pipe = trk.stage_ms()
        trk.enable_profiling(0)
        base += 9
        step(base, announce_next=False)   # consumes the read-ahead that is still outstanding
        base += 1
    plan_hits, plan_misses = trk.plan_stats()
    trk.enable_profiling(2)
    for i in range(base, base + 8):
        step(i, announce_next=False)  # serial frames: every stage on the main stream so that each has a duration
    stage_all = trk.stage_ms()
    trk.enable_profiling(0)
    # U (voxels updated) and S (ray-march samples) of exactly the timed frames: the run is deterministic (and identical with and
    # without read-ahead), so the same frames are replayed from a reset tracker with the counting kernel variants
    trk.reset()
    trk.enable_counts(True)
    Us, Ss, Ls = [], [], []
    for i in range(args.warmup + args.steps):
        step(i, announce_next=False)
        if i >= args.warmup:
            U, S = trk.last_counts()
            Us.append(U)
            Ss.append(S)
            if len(Ls) < 16:   # lane-steps of the tsdf23 launch = wave batches x 4 z-steps x 64 lanes (diagnostic; drains the GPU)
                try:
                    ctx.sync()
                    dc = trk.debug_counts()
                    if dc[0] > 0:
                        Ls.append((dc[0], dc[1] * 4 * 64))
                except Exception as e:  # noqa: BLE001 -- a diagnostic must not cost the bench line
                    sys.stderr.write(f"bench: lane-step diagnostic unavailable ({e})\n")
                    Ls = [None] * 16
    trk.enable_counts(False)
    if trk.num_poses() != args.warmup + args.steps:
        sys.stderr.write(f"bench: the counting replay produced {trk.num_poses()} poses for {args.warmup + args.steps} frames\n")
    # A/B of the voxel kernel's two contracts on this workload (untimed, alone, serial frames; after every other measurement: the volume the
    # tolerant kernel leaves behind is not used again).  Never the headline: `roofline` above is the contract the timed frames ran.
    contract_ab = None
    if abi.lib().kt_debug_tsdf_kernel().decode() == "kt_tsdf23_lean_kernel" and rank == 0 and world == 1 and args.warmup + args.steps >= 20 and not args.no_contract_ab:
        n_ab = min(args.warmup + args.steps, 60)
        def alone_ms(contract):   # 0 bit-exact, 1 survey-8c, 2 the speed-of-light measurement variant (kt_tsdf23_sol_kernel)
            abi._chk(abi.lib().kt_debug_tsdf_contract(int(contract)))
            trk.reset()
            for i in range(n_ab - 16):
                step(i, announce_next=False)
            trk.pose()
            trk.enable_profiling(2)
            for i in range(n_ab - 16, n_ab):
                step(i, announce_next=False)
            trk.pose()
            ms_ = trk.stage_ms()["tsdf23"][0]
            trk.enable_profiling(0)
            return ms_
        try:
            ab = [alone_ms(0), alone_ms(1), alone_ms(2), alone_ms(0), alone_ms(1), alone_ms(2)]
        finally:
            abi._chk(abi.lib().kt_debug_tsdf_contract(-1))
        contract_ab = {"frames": [n_ab - 16, n_ab], "bit_exact_alone_ms": [round(ab[0], 5), round(ab[3], 5)], "survey8c_alone_ms": [round(ab[1], 5), round(ab[4], 5)],
                       # NOT a contract: the per-voxel arithmetic the reference's --prec-div=false --prec-sqrt=false build executes, same loads / stores / predicates /
                       # task list (kt_tsdf23_sol_kernel); what no arithmetic contract of this decomposition can beat
                       "speed_of_light_alone_ms": [round(ab[2], 5), round(ab[5], 5)]}
    U = float(np.mean(Us))
    Lok = [x for x in Ls if x]
    lane_eff = round(sum(u for u, _ in Lok) / max(1, sum(l for _, l in Lok)), 4) if Lok else None
    P = cam.cols * cam.rows
    # algorithmic bytes of the tsdf23 launch (DESIGN.md "integrate"): 12 B per updated voxel (2 B tsdf + 4 B colour/weight,
    # read and written) + the per-pixel record gathered by the voxels (12 B, counted once per pixel)
    bytes_tsdf23 = 12.0 * U + 12.0 * P
    if tsdf23_n == 0 or tsdf23_ms <= 0:   # very short runs: no timed frame carried the event pair -> the untimed stage pass's launches
        tsdf23_ms, tsdf23_n = stage_all["tsdf23"][0], 0
    achieved = bytes_tsdf23 / (tsdf23_ms * 1e-3) / 1e9 if tsdf23_ms > 0 else 0.0
    peak = 8000.0
    # HBM-side traffic of the same kernel: PMC counters cannot be read from inside the process, so this is the committed rocprofv3
    # measurement of this workload (profiles/, collected and corrected as MI355X_MICROARCH.md prescribes: separate --pmc passes,
    # FETCH_SIZE x 2 after calibration on this access width, WRITE_SIZE x 1, KiB units); null for workloads without one
    traffic, traffic_ratio, traffic_source = committed_traffic(args.workload) if N == WORKLOADS[args.workload][2] else (None, None, None)
    voxel_kernel = abi.lib().kt_debug_tsdf_kernel().decode()

    out = {
        "metric": "RGB-D frames/sec @640x480, 512^3 TSDF" if args.workload == "orbit512" else f"RGB-D frames/sec ({args.workload})",
        "value": world * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {cam.cols}x{cam.rows} synthetic {cfg_name} sequence, {'ICP+RGB-D' if d['use_rgbd_icp'] else 'ICP-only'} tracking, "
                               f"{N}^3 TSDF, " + ("inputs in HOST memory (PCIe-inclusive)" if args.host_frames else "inputs resident in HBM") + ", 1 stream per GPU (" + BASELINE_CONFIG[args.workload] + ")"
                               + (", log playback with 1 frame of read-ahead" if readahead else ", no read-ahead"),
                   "volume": N, "cols": cam.cols, "rows": cam.rows, "unique_frames": nuniq, "pose_err_m_at_end": pose_err,
                   "pose_gather_bytes": pose_bytes,
                   # the side streams wait for the ray cast of the frame in flight instead of running beside its voxel kernel (KT_SIDE_GATE;
                   # default: dense views only); frames whose odometry was re-run stepwise after a hand-off time-out (0 on an undisturbed GPU)
                   "side_gate": int(abi.lib().kt_tracker_debug_side_gate(trk.h)), "odometry_fallbacks": fallbacks,
                   "frame_ms": {"p50": round(float(np.percentile(periods, 50)), 4), "p99": round(float(np.percentile(periods, 99)), 4),
                                "max": round(float(periods.max()), 4),
                                # the slowest calls (index in the timed region, ms): shift frames and whatever else stalls the caller
                                "slowest": [[int(i), round(float(periods[i]), 3)] for i in np.argsort(periods)[::-1][:4]]},
                   "slices_by_direction": slices_by_dim, **({"periods_ms": [round(float(x), 3) for x in periods]} if os.environ.get("KT_BENCH_PERIODS") else {})},
        "roofline": {"kernel": voxel_kernel, "contract": "survey-8c" if voxel_kernel == "kt_tsdf23_tol_kernel" else "bit-exact", "bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                     "frac": achieved / peak,
                     "frac_is": "the launch inside the timed region, next to the read-ahead and plan streams (since round 5 they run beside the fusion kernels: the "
                                "odometry of a frame is one resident launch that takes every compute unit); frac_alone = the same launch with nothing else on the GPU",
                     "traffic": traffic, "traffic_ratio": traffic_ratio,
                     # not measured by this run: the committed rocprofv3 PMC passes of the same workload and the same kt_volume.hip
                     "traffic_source": traffic_source, "algorithmic_bytes_per_launch": bytes_tsdf23,
                     "avg_launch_ms": tsdf23_ms, "launches_timed": tsdf23_n, "U_voxels_updated": U, "S_raycast_steps": float(np.mean(Ss)),
                     # lanes the launch spends per updated voxel (first 16 timed frames): 64-lane wave z-steps over wave-columns of 32 x 2
                     # voxel columns, profiles/r02_tsdf23_whatif.md
                     "lane_efficiency": lane_eff,
                     # the same launch (alone, frames [a, b) of the sequence) under the two contracts, alternating; informational
                     "contract_ab": contract_ab,
                     # in the timed region the kernel shares the GPU with the read-ahead stream (next frame's bilateral / pyramid);
                     # the same launch with nothing else running (untimed stage pass below):
                     "avg_launch_ms_alone": stage_all["tsdf23"][0],
                     "frac_alone": (bytes_tsdf23 / (stage_all["tsdf23"][0] * 1e-3) / 1e9 / peak) if stage_all["tsdf23"][0] > 0 else None,
                     # the whole integrate stage of the pipelined frame by SURVEY 8(d)'s stage formula (12 U + 25 P: the voxel words, the raw
                     # depth, the scaled depth and the pixel records written and read once) over the stage's time on the main stream
                     "stage_frac": (((12.0 * U + 25.0 * cam.cols * cam.rows) / (stage_pipe["integrate"][0] * 1e-3) / 1e9 / peak)
                                    if stage_pipe and stage_pipe.get("integrate", (0,))[0] > 0 else None)},
        # every stage alone on the main stream (serial frames, in-stream pre-pass); stage_ms_pipelined: the main stream's stages with the
        # read-ahead stream running and the voxel pass planned ahead, i.e. as in the timed region (pyramid / resize are not on it there)
        "stage_ms": {k: round(v[0], 4) for k, v in stage_all.items()},
        "stage_ms_pipelined": ({k: round(v[0], 4) for k, v in stage_pipe.items() if k in ("odometry", "shift", "integrate", "raycast", "tsdf23")} if stage_pipe else None),
        "planned_frames": {"hits": plan_hits, "misses": plan_misses},
        "host_ms_per_frame": {"process_frame_call": round(1e3 * host_call_s, 4), "of_which_waiting_for_pose": round(1e3 * host_wait_s, 4)},
    }

    if rank == 0 and world == 1 and args.workload == "orbit512" and not args.no_stress and not args.host_frames:
        out["roofline_stress"] = roofline_stress(ctx, abi, synth)
        out["roofline_stress"]["kernel"] = voxel_kernel

    # every rank checks its row of the gather and leaves the communicator TOGETHER, before rank 0 alone spends seconds on the CPU
    # baseline (a rank destroying its communicator while a peer still holds one is the first thing to hang on real hardware)
    if comm is not None:
        check_gather(allp, rank, np.stack([p.reshape(16) for p in timed_poses]))
        with stdout_to_stderr():
            comm.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # (rank 0 at N = 1 only: the contract's cpu_baseline leg)
        # the oracle needs cpu_frames + 2 frames of the same sequence whatever --steps / --warmup are
        cframes = frames if len(frames) >= args.cpu_frames + 2 else synth.sequence(cfg_name, args.cpu_frames + 2, cam, seed)[1]
        out["cpu_baseline"] = cpu_baseline(cam, N, d, cframes, args.cpu_frames, first_poses)

    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    os.dup2(2, 1)   # whatever the libraries still print while shutting down does not reach the driver's stdout
    trk.close()
    ctx.close()


def kt_volume_sha():
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "kintinuous_amd", "csrc", "kt_volume.hip"), "rb").read()).hexdigest()[:16]


def committed_traffic(workload):
    """HBM-side traffic of the tsdf23 launch: PMC counters cannot be read from inside the process, so this is the committed rocprofv3
    measurement of this workload (profiles/r05_pmc_tsdf23_<workload>.json, written by scripts/pmc_traffic.sh: separate --pmc passes,
    FETCH_SIZE x 2 after calibration on the kernel's own access pattern, WRITE_SIZE x 1, as MI355X_MICROARCH.md prescribes).  It is only
    quoted for the kernel it was measured on: the file records the sha256 of kt_volume.hip, and a stale measurement prints null.
    Returns (bytes per launch, bytes per launch / algorithmic bytes of the SAME launches, the file): the profiled run covers other frames than
    the timed region, so the ratio -- not the absolute -- is what compares with this line's algorithmic bytes."""
    for rnd in ("r06", "r05", "r04", "r03"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_pmc_tsdf23_{workload}.json")
        try:
            j = json.load(open(f))
            if j.get("kt_volume_hip_sha16") != kt_volume_sha():
                continue
            return float(j["traffic_bytes_per_launch"]), (float(j["traffic_ratio"]) if "traffic_ratio" in j else None), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None, None


def roofline_stress(ctx, abi, synth):
    """BASELINE.json configs[4] / SURVEY 8(d) config 5, the roofline showcase: 1280x960 depth into a 768^3 volume in static mode (the
    camera 0.45 m outside the near face, far wall at 6.2 m), 8 warm-up + 4 timed frames, HIP events around every tsdf23 launch.
    Three measurements of the SAME launch, each labelled (the headline `roofline.frac` is an in-region figure, this
    block's `frac` was an alone figure and did not say so):
      frac_alone / avg_launch_ms_alone : frames pushed one at a time, nothing else on the GPU (`frac` = this one, as in rounds 2-4);
      frac_pipelined / avg_launch_ms_pipelined : frames pushed with one frame of read-ahead, as the headline's timed region runs them
          (the 1280x960 bilateral / pyramid / scaleDepth of the NEXT frame share the GPU with the voxel kernel);
      survey8c : the same two under the voxel kernel's second contract (kt_tsdf23_tol_kernel) -- an A/B of the contracts, never the headline."""
    N, cam = 768, synth.Camera.scaled(2)
    _, frames, _, kw = synth.sequence("farwall", 5, cam)
    cfg = abi.TrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, 6.0, 14, 2, 1, 0, 0, 0, 0, 0)
    trk = abi.Tracker(ctx, cfg)
    dev = [(ctx.upload(a), ctx.upload(b)) for a, b in frames]
    WARM, TIMED = 8, 4   # (round 4: 8 warm-up frames instead of 2 -- the first launches into the freshly cleared 2.7 GB of volumes run 8 % slower)
    seq = [pingpong(k, len(dev)) for k in range(WARM + 16)]

    def run(readahead, timed=TIMED):
        """-> (mean tsdf23 launch ms, launches, frame ms) over `timed` frames after WARM; the tracker is reset first.  A frame whose plan is
        rejected is parked: the event pair then sits around an EMPTY launch (3.5 us) and the real one, re-issued by the host, is not
        timed -- such pairs (plan_stats misses inside the window) are taken out of the mean instead of flattering it."""
        TIMED = timed
        trk.reset()
        push = lambda k: trk.process_frame(dev[seq[k]][0], dev[seq[k]][1], 33333 * k)
        ahead = lambda k: trk.prefetch_frame(dev[seq[k]][0], dev[seq[k]][1])
        for k in range(WARM):
            if readahead:
                ahead(k + 1)
            push(k)
        trk.pose()
        ctx.sync()
        misses0 = trk.plan_stats()[1]
        trk.enable_profiling(4)
        t0 = time.perf_counter()
        for k in range(WARM, WARM + TIMED):
            if readahead:
                ahead(k + 1)
            push(k)
        trk.pose()
        ctx.sync()
        frame_ms = 1e3 * (time.perf_counter() - t0) / TIMED
        if readahead:
            ahead(WARM + TIMED + 1)
        push(WARM + TIMED)   # harvest the last event pair
        trk.pose()
        ctx.sync()
        ms, n = trk.stage_ms()["tsdf23"]
        trk.enable_profiling(0)
        parked = trk.plan_stats()[1] - misses0
        if 0 < parked < n:
            ms, n = (ms * n - parked * 0.0035) / (n - parked), n - parked
        return ms, n, frame_ms

    side_gate = int(abi.lib().kt_tracker_debug_side_gate(trk.h))
    ms, n, frame_ms = run(False)
    ms_p, n_p, frame_ms_p = run(True, 12)
    tol = sol = None
    if abi.lib().kt_debug_tsdf_kernel().decode() == "kt_tsdf23_lean_kernel":   # the A/B of the contracts (skipped when the run itself is under survey-8c)
        abi._chk(abi.lib().kt_debug_tsdf_contract(1))
        try:
            tol = (run(False), run(True, 12))
            abi._chk(abi.lib().kt_debug_tsdf_contract(2))
            sol = run(False)
        finally:
            abi._chk(abi.lib().kt_debug_tsdf_contract(-1))
    trk.reset()
    trk.enable_counts(True)
    Us = []
    for k in range(WARM + TIMED):   # U of exactly the timed frames: the same frames replayed with the counting kernel
        trk.process_frame(dev[seq[k]][0], dev[seq[k]][1], 33333 * k)
        if k >= WARM:
            Us.append(trk.last_counts()[0])
    trk.close()
    U, P = float(np.mean(Us)), cam.cols * cam.rows
    b = 12.0 * U + 12.0 * P
    frac = lambda t: (b / (t * 1e-3) / 1e9 / 8000.0) if t > 0 else None
    achieved = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = committed_traffic("farwall768")
    out = {"workload": "farwall768: 1280x960 synthetic far-wall sequence, static mode, 768^3 TSDF (BASELINE.json configs[4])", "kernel": "kt_tsdf23_kernel",
           "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
           "frac_is": "alone: frames one at a time, no read-ahead stream next to the launch (frac_pipelined = with it, as the headline's region)",
           "frac_alone": frac(ms), "avg_launch_ms_alone": ms,
           "frac_pipelined": frac(ms_p), "avg_launch_ms_pipelined": ms_p, "launches_timed_pipelined": n_p, "frame_ms_pipelined": round(frame_ms_p, 3),
           "traffic": traffic[0], "traffic_ratio": traffic[1], "traffic_source": traffic[2],
           "algorithmic_bytes_per_launch": b, "avg_launch_ms": ms, "launches_timed": n, "U_voxels_updated": U, "frame_ms": round(frame_ms, 3),
           "side_gate": side_gate}
    if sol:
        # the falsifiable form of "the exactness contract is not what keeps this launch from 0.60" (VERDICT r5 item 1d): the SAME launch -- task list,
        # loads, stores, predicates -- with the per-voxel arithmetic of the reference's --prec-div=false --prec-sqrt=false build (kt_tsdf23_sol_kernel)
        out["speed_of_light"] = {"kernel": "kt_tsdf23_sol_kernel", "frac_alone": frac(sol[0]), "avg_launch_ms_alone": sol[0],
                                 "note": "measurement variant, results are not the reference's; never the headline"}
    if tol:
        out["survey8c"] = {"kernel": "kt_tsdf23_tol_kernel", "frac_alone": frac(tol[0][0]), "avg_launch_ms_alone": tol[0][0],
                           "frac_pipelined": frac(tol[1][0]), "avg_launch_ms_pipelined": tol[1][0]}
    return out


def cpu_baseline(cam, N, d, frames, nframes, gpu_poses=()):
    """The oracle (CPU restatement of the reference -- the reference has no CPU path) timed on this host's cores on the
    first `nframes` frames of the same workload.  A reported baseline, not the target."""
    cores = min(32, len(os.sched_getaffinity(0)))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle import oracle
    oracle.build()
    ocfg = oracle.OTrackerConfig(cam.cols, cam.rows, N, cam.fx, cam.fy, cam.cx, cam.cy, d["volume_size"], d["voxel_shift"], d["overlap"],
                                 d["static_mode"], d["use_rgbd"], d["use_rgbd_icp"], d["fast_odometry"], d["disable_color_angle"], 0)
    otr = oracle.OracleTracker(ocfg)
    n = min(nframes, len(frames))
    otr.process_frame(frames[0][0], frames[0][1], 0)  # frame 0 only integrates; not representative, excluded
    t0 = time.perf_counter()
    for k in range(1, n):
        otr.process_frame(frames[k][0], frames[k][1], 33333 * k)
    dt = time.perf_counter() - t0
    # the run is paid for: also compare its poses with the GPU's on the same frames (the parity tests proper are tests/test_gpu_configs.py)
    compared = min(n, len(gpu_poses))
    pose_diff = max((float(np.abs(otr.dense_pose(k)[1] - gpu_poses[k]).max()) for k in range(compared)), default=None)
    stages = otr.stage_seconds()
    # the same tracker on ONE thread, two more frames (SURVEY 8d asks for both ends of the host's range)
    single = None
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        t1 = time.perf_counter()
        m = 0
        for k in range(n, min(n + 2, len(frames))):
            otr.process_frame(frames[k][0], frames[k][1], 33333 * k)
            m += 1
        if m:
            single = m / (time.perf_counter() - t1)
        gomp.omp_set_num_threads(cores)
    except OSError:
        pass
    otr.close()
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": (n - 1) / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"frames 1..{n - 1} of the same sequence through oracle/ (OpenMP, {cores} threads) on {model}",
            "stage_s_total": {k: round(v, 3) for k, v in stages.items()},
            "single_thread_value": single, "pose_max_abs_diff_vs_gpu": pose_diff, "poses_compared": compared}


if __name__ == "__main__":
    main()
