"""GPU: two PROCESSES on device 0, both with the library's defaults.  kt_icp_level_kernel needs its whole grid resident and
its workgroups wait for each other inside the launch; a process cannot see another process's kernels (kt_live_trackers is per process), so two
level launches can each hold part of the machine.  The waits are bounded in time, a launch that gives up aborts, the frame's odometry is re-run in
the stepwise form and the level form stays off for a while: both processes must finish every frame with NO error and with the poses of a
solo run, bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "tools", "two_process_worker.py")


def _spawn(args, extra_env=None):
    env = dict(os.environ)
    env.pop("KT_ICP_LEVELS", None)   # both default settings
    env.pop("KT_PREPARE_FUSED", None)
    env.update(extra_env or {})
    return subprocess.Popen([sys.executable, WORKER] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)


def _result(p):
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, out + err
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_two_processes_share_one_gpu(tmp_path):
    solo = _result(_spawn(["--frames", "30", "--passes", "1"]))
    assert solo["fallbacks"] == 0 and solo["level_form_last"] == 1   # alone, the level form runs undisturbed
    ps = [_spawn(["--frames", "30", "--passes", "3", "--barrier", str(tmp_path), "--me", str(k), "--peers", "2"]) for k in range(2)]
    res = [_result(p) for p in ps]
    for r in res:
        for one_pass in r["poses"]:
            assert one_pass == solo["poses"][0]      # bit-equal to the solo run, every pass, both processes
    print("two processes on one GPU: fallbacks %s, seconds %s (solo %.2f for one pass)" % ([r["fallbacks"] for r in res], [round(r["seconds"], 2) for r in res], solo["seconds"]))


def test_fused_frame_preparation_equals_the_separate_launches():
    """Round 6: the tracker prepares a frame with kt_frame_prepare (kt_pyramid01_kernel, then pyramid levels 2 / 3 and scaleDepth + pixel records
    in ONE launch) instead of kt_build_pyramid + kt_integrate_prepare.  The switch is read once per process, hence two processes: the same
    sequence with KT_PREPARE_FUSED=0 and with the default must leave the same poses and the same volumes, bit for bit.  (Both forms are compared
    with the oracle elsewhere -- the tracker tests run the fused form, test_gpu_image.py / test_gpu_volume.py the stand-alone entry points.)"""
    a = _result(_spawn(["--frames", "24", "--passes", "1"]))
    b = _result(_spawn(["--frames", "24", "--passes", "1"], {"KT_PREPARE_FUSED": "0"}))
    assert a["poses"] == b["poses"] and a["volumes_sha256"] == b["volumes_sha256"]
