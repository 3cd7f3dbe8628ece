"""Fuzzes the untrusted-input parsers of the host side under AddressSanitizer + UBSan: the JPEG decoder (host/JpegDecoder.h via
jpeg_tool) and the .klg reader (host/RawLogReader.h via klg_tool, synchronous and with decode-ahead threads, which must agree).  Mutated and truncated streams must either decode or be rejected
with exit code 1 -- never crash.  Not part of the pytest run (minutes);   python tests/tools/fuzz_host_io.py [iterations]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kintinuous_amd import jpeg_ref, klg, synth  # noqa: E402

HOST = os.path.join(ROOT, "kintinuous_amd", "host")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    tmp = tempfile.mkdtemp()
    jt, kt = os.path.join(tmp, "jpeg_tool_san"), os.path.join(tmp, "klg_tool_san")
    subprocess.check_call(["g++"] + SAN + [os.path.join(HOST, "jpeg_tool.cpp"), "-o", jt])
    subprocess.check_call(["g++"] + SAN + [os.path.join(HOST, "klg_tool.cpp"), "-o", kt, "-lz", "-pthread"])
    rng = np.random.default_rng(123)
    yy, xx = np.mgrid[0:50, 0:67]
    img = np.stack([(xx * 3) % 256, (yy * 5) % 256, (xx + yy) % 256], -1).astype(np.uint8)
    streams = [jpeg_ref.encode(img, subsampling=s, restart_interval=r, ac_table=t, interleaved=i)
               for s, r, t, i in (("420", 0, "flat", True), ("422", 2, "skewed", True), ("444", 0, "flat", False))] + [jpeg_ref.encode(img[..., 0].copy())]
    cam = synth.Camera.small(64, 48)
    frames = [synth.render(synth.Scene("room"), cam, *p) for p in synth.orbit_trajectory(3)]
    logs = []
    for kw in (dict(), dict(compress_depth=True), dict(compress_depth=True, jpeg_quality=80)):
        p = os.path.join(tmp, "b.klg")
        klg.write_klg(p, frames, cols=64, rows=48, **kw)
        logs.append(open(p, "rb").read())

    def mutate(s, k, head=None):
        s = bytearray(s)
        if k % 3 == 0:
            for _ in range(rng.integers(1, 6)):
                s[rng.integers(0, head or len(s))] = rng.integers(0, 256)
        elif k % 3 == 1:
            s = s[:rng.integers(4, len(s))]
        else:
            a, n = rng.integers(2, len(s) - 8), rng.integers(1, 8)
            s[a:a + n] = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        return bytes(s)

    crashes = 0
    for k in range(iters):
        f = os.path.join(tmp, "x.jpg")
        open(f, "wb").write(mutate(streams[k % len(streams)], k))
        r = subprocess.run([jt, f, "67", "50", os.path.join(tmp, "x.bgr")], capture_output=True, text=True, timeout=60)
        if r.returncode not in (0, 1):
            crashes += 1
            print("JPEG CRASH", k, r.returncode, r.stderr[-800:])
        f = os.path.join(tmp, "x.klg")
        open(f, "wb").write(mutate(logs[k % 3], k, head=200 if k % 6 == 0 else None))
        r = subprocess.run([kt, "-l", f, "-w", "64", "-h", "48"], capture_output=True, text=True, timeout=60)
        if r.returncode not in (0, 1):
            crashes += 1
            print("KLG CRASH", k, r.returncode, r.stderr[-800:])
        # the decode-ahead reader on the same damaged log: same frames, same verdict, same message
        r2 = subprocess.run([kt, "-l", f, "-w", "64", "-h", "48", "-dt", str(1 + k % 5)], capture_output=True, text=True, timeout=60)
        if (r2.returncode, r2.stdout, r2.stderr) != (r.returncode, r.stdout, r.stderr):
            crashes += 1
            print("KLG DECODE-AHEAD DIFFERS", k, r.returncode, r2.returncode, r.stderr[-300:], r2.stderr[-800:])
    print("iterations", iters, "crashes", crashes)
    return 1 if crashes else 0


if __name__ == "__main__":
    sys.exit(main())
