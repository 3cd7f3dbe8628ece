""".klg log I/O (reference: src/utils/RawLogReader.cpp:21-150).

Layout: int32 numFrames; per frame int64 timestamp_us, int32 depthSize, int32 imageSize, depth bytes, image bytes.
Depth is raw little-endian uint16[W*H] when depthSize == 2*W*H, else a zlib stream of the same.  Image is raw
uint8[W*H*3] when imageSize == 3*W*H, all zero when imageSize == 0; a JPEG payload (0 < imageSize < 3*W*H) needs a
decoder that this environment does not have and raises.  Like the reference reader, iteration stops one frame
early: hasMore() is currentFrame + 1 < numFrames (RawLogReader.cpp:147-150).
"""
from __future__ import annotations

import struct
import zlib
from typing import Iterator, Tuple

import numpy as np


def write_klg(path: str, frames, timestamps=None, cols: int = 640, rows: int = 480, compress_depth: bool = False,
              jpeg_quality: int = 0) -> None:
    """jpeg_quality > 0: the colour image is stored as a baseline JPEG (what the reference's Logger2 records; needs zlib depth,
    RawLogReader.cpp:99-108), encoded by kintinuous_amd/jpeg_ref.py."""
    assert not jpeg_quality or compress_depth
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for k, (depth, rgb) in enumerate(frames):
            ts = timestamps[k] if timestamps is not None else 33333 * k
            d = np.ascontiguousarray(depth, dtype="<u2").tobytes()
            img = np.ascontiguousarray(rgb, dtype=np.uint8).tobytes()
            assert len(d) == 2 * cols * rows and len(img) == 3 * cols * rows
            if compress_depth:
                d = zlib.compress(d)
            if jpeg_quality:
                from . import jpeg_ref
                img = jpeg_ref.encode(np.ascontiguousarray(rgb, dtype=np.uint8).reshape(rows, cols, 3), quality=jpeg_quality)
            f.write(struct.pack("<qii", ts, len(d), len(img)))
            f.write(d)
            f.write(img)


def read_klg(path: str, cols: int = 640, rows: int = 480, reference_quirk: bool = True) -> Iterator[Tuple[int, np.ndarray, np.ndarray]]:
    """Yields (timestamp, depth[rows, cols] uint16, rgb[rows, cols, 3] uint8).  With reference_quirk the last frame of
    the log is never produced, exactly like RawLogReader::hasMore()."""
    P = cols * rows
    with open(path, "rb") as f:
        (n,) = struct.unpack("<i", f.read(4))
        last = n - 1 if reference_quirk else n
        for k in range(n):
            ts, dsz, isz = struct.unpack("<qii", f.read(16))
            d = f.read(dsz)
            img = f.read(isz) if isz > 0 else b""
            if k >= last:
                break
            if dsz == 2 * P:
                depth = np.frombuffer(d, dtype="<u2").reshape(rows, cols)
            elif dsz > 0:
                depth = np.frombuffer(zlib.decompress(d), dtype="<u2").reshape(rows, cols)
            else:
                depth = np.zeros((rows, cols), np.uint16)
            if isz == 3 * P:
                rgb = np.frombuffer(img, dtype=np.uint8).reshape(rows, cols, 3)
            elif isz == 0:
                rgb = np.zeros((rows, cols, 3), np.uint8)
            else:  # cvDecodeImage in the reference: B G R bytes of the JPEG stream
                from . import jpeg_ref
                rgb = jpeg_ref.decode(img)
                assert rgb.shape == (rows, cols, 3)
            yield ts, depth.copy(), rgb.copy()


def write_poses(path: str, poses) -> None:
    """<log>.poses as KintinuousTracker::outputPose writes it (KintinuousTracker.cpp:199-218): '%.6f tx ty tz qx qy qz qw'
    with the default 6-significant-digit ostream formatting of the seven floats."""
    with open(path, "w") as f:
        for ts, t, q in poses:
            f.write("%.6f " % (ts / 1000000.0))
            f.write(" ".join("%g" % v for v in list(t) + list(q)) + "\n")


PCD_NORMAL_DTYPE = np.dtype([("xyz", np.float32, 3), ("bgra", np.uint8, 4), ("normal", np.float32, 3), ("curvature", np.float32)])


def read_pcd(path: str) -> np.ndarray:
    """Binary PCD as `kintinuous_hip` writes it: `-pcdraw` (FIELDS x y z rgb -> {xyz, bgra}) or `-pcd`, the reference's saved cloud
    (pcl::PointXYZRGBNormal: FIELDS x y z rgb normal_x normal_y normal_z curvature -> PCD_NORMAL_DTYPE)."""
    with open(path, "rb") as f:
        n, dt = None, None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("FIELDS"):
                fields = line.split()[1:]
                if fields == ["x", "y", "z", "rgb"]:
                    dt = np.dtype([("xyz", np.float32, 3), ("bgra", np.uint8, 4)])
                else:
                    assert fields == ["x", "y", "z", "rgb", "normal_x", "normal_y", "normal_z", "curvature"], line
                    dt = PCD_NORMAL_DTYPE
            if line.startswith("POINTS"):
                n = int(line.split()[1])
            if line.startswith("DATA"):
                assert line.split()[1] == "binary", line
                break
        return np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
