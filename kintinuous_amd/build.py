"""Builds libkt_hip.so (the HIP kernels + C-ABI) in-tree with hipcc for gfx950.

No CPU fallback exists: if hipcc is missing this raises.  The .so is git-ignored but travels with the
gpurun snapshot, so GPU boxes use the file built here.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libkt_hip.so")
BUILD_DIR = os.path.join(os.path.dirname(_HERE), "build")
SOURCES = ["kt_context.hip", "kt_image.hip", "kt_volume.hip", "kt_track.hip", "kt_tracker.hip", "kt_hostmath.hip", "kt_comm.hip", "kt_slice.hip", "kt_cloud.hip"]
# measurement kernels (PMC calibration streams, instruction issue rates, the exhaustive division check): a library of their own, loaded by
# scripts/ and one test -- the product library carries none of them
DEBUG_OUT = os.path.join(_HERE, "libkt_debug.so")
DEBUG_SOURCES = ["kt_debug.hip"]
# -ffp-contract=off: a*b+c fuses only where __builtin_fmaf is written (bit-parity with the oracle);
# IEEE division / sqrt are hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libkt_hip.so cannot be built (there is no CPU fallback)")


def _deps():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "kt_abi.h"))
    return files


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(DEBUG_OUT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(DEBUG_OUT))
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cc = hipcc()
    os.makedirs(BUILD_DIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(BUILD_DIR, os.path.splitext(src)[0] + ".o")
        cmd = [cc] + FLAGS + os.environ.get("KT_EXTRA_FLAGS", "").split() + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES) + len(DEBUG_SOURCES)) as ex:
        allobjs = list(ex.map(compile_one, SOURCES + DEBUG_SOURCES))
    objs, dobjs = allobjs[:len(SOURCES)], allobjs[len(SOURCES):]
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", DEBUG_OUT] + dobjs + ["-L", _HERE, "-lkt_hip", "-Wl,-rpath,$ORIGIN"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link of libkt_debug.so failed:\n{r.stderr}")
    return OUT


HOST_DIR = os.path.join(_HERE, "host")
HOST_BIN = os.path.join(HOST_DIR, "bin", "kintinuous_hip")
JPEG_TOOL = os.path.join(HOST_DIR, "bin", "jpeg_tool")
KLG_TOOL = os.path.join(HOST_DIR, "bin", "klg_tool")
CONSUMER_TEST = os.path.join(HOST_DIR, "bin", "consumer_test")
CONTROLLER_TEST = os.path.join(HOST_DIR, "bin", "controller_test")


def build_host(force: bool = False) -> str:
    """g++ build of the C++ host shell's headless driver (kintinuous_amd/host/main.cpp) against libkt_hip.so."""
    deps = [os.path.join(dp, f) for dp, _, fs in os.walk(HOST_DIR) for f in fs if f.endswith((".h", ".hpp", ".cpp"))]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "kt_abi.h"))
    outs = (HOST_BIN, JPEG_TOOL, KLG_TOOL, CONSUMER_TEST, CONTROLLER_TEST)
    if not force and all(os.path.exists(o) for o in outs) and all(os.path.getmtime(d) <= min(os.path.getmtime(o) for o in outs) for d in deps):
        return HOST_BIN
    os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
    # the .klg colour decoder on its own (no HIP dependency): used by the CPU tests
    for src, out, libs in (("jpeg_tool.cpp", JPEG_TOOL, ["-lz"]), ("klg_tool.cpp", KLG_TOOL, ["-lz", "-pthread"])):
        r = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", os.path.join(HOST_DIR, src), "-o", out] + libs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{src} build failed:\n{r.stderr}")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.dirname(_HERE), os.path.join(HOST_DIR, "main.cpp"), "-o", HOST_BIN,
           "-L", _HERE, "-lkt_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/../..", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host shell build failed:\n{r.stderr}")
    # the backend-consumer stand-in (CloudSliceProcessor's tracker-facing half) against the same shell: compiling it is half the test
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", os.path.dirname(_HERE), os.path.join(HOST_DIR, "consumer_test.cpp"), "-o", CONSUMER_TEST,
           "-L", _HERE, "-lkt_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/../..", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"consumer_test build failed:\n{r.stderr}")
    # MainController minus the GUI (setup / mainLoop / complete / save) against the shell's ThreadObjects: compiling it is half the test
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", os.path.dirname(_HERE), os.path.join(HOST_DIR, "controller_test.cpp"), "-o", CONTROLLER_TEST,
           "-L", _HERE, "-lkt_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/../..", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"controller_test build failed:\n{r.stderr}")
    return HOST_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv))
