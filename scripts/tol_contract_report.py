"""What the survey-8c contract of the voxel kernel (kt_tsdf23_tol_kernel) changes, COUNTED against the oracle at BASELINE configs 2 / 3 / 5
(tests/test_gpu_tol.py::report) -> one JSON line per config.   gpurun: python scripts/tol_contract_report.py > gpurun_out/r05_tol_contract.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(32, len(os.sched_getaffinity(0)))))

from kintinuous_amd import abi, synth  # noqa: E402
from oracle import oracle  # noqa: E402
import test_gpu_tol as T  # noqa: E402

oracle.build()
ctx = abi.Ctx(0)
cam2 = synth.Camera(1280, 960, 2 * synth.FX, 2 * synth.FY, 2 * synth.CX, 2 * synth.CY)
for cfg, N, grow, cam, wrap in (("orbit", 512, 12, None, (37, 501, 130)), ("crabwalk", 512, 12, None, (5, 0, 500)), ("farwall", 768, 1, cam2, (0, 0, 0)),
                                ("orbit", 512, 60, None, (0, 0, 0))):
    print(json.dumps({"kind": "one call on identical inputs", **T.kernel_report(ctx, cfg, N, grow, cam, wrap)}), flush=True)
for args in (("orbit", 34, 512), ("crabwalk", 29, 512), ("orbit", 120, 512)):
    print(json.dumps({"kind": "whole run vs the oracle", **T.report(ctx, *args)}), flush=True)
ctx.close()
