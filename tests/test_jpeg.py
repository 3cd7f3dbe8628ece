"""CPU: the .klg colour decoder (kintinuous_amd/host/JpegDecoder.h, the stand-in for cvDecodeImage in RawLogReader) against an
independent numpy restatement of libjpeg's default decode path (kintinuous_amd/jpeg_ref.py): byte-identical output for every
stream layout the encoder can produce; the integer IDCT path within 2 grey levels of a double-precision decode; decent PSNR
against the source image.  Pinned against the real library as well: Pillow (libjpeg-turbo, the decoder OpenCV's cvDecodeImage
wraps too) decodes our encoder's streams AND its own to exactly the bytes the C++ decoder produces."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool():
    from kintinuous_amd import build
    build.build_host()
    assert os.path.exists(build.JPEG_TOOL)
    return build.JPEG_TOOL


def _cxx_decode(tool, data, w, h, tmp_path, name="x"):
    src, dst = tmp_path / f"{name}.jpg", tmp_path / f"{name}.bgr"
    src.write_bytes(data)
    r = subprocess.run([tool, str(src), str(w), str(h), str(dst)], capture_output=True, text=True, timeout=60)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return np.frombuffer(dst.read_bytes(), np.uint8).reshape(h, w, 3)


def _images():
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    _, rgb = synth.render(synth.Scene("room"), cam, *synth.orbit_trajectory(2)[0])
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (23, 37, 3), dtype=np.uint8)            # ragged size, worst-case content
    yy, xx = np.mgrid[0:50, 0:67]
    ramp = np.stack([(xx * 3) % 256, (yy * 5) % 256, (xx + yy) % 256], -1).astype(np.uint8)
    return {"render": np.ascontiguousarray(rgb), "noise": noise, "ramp": ramp}


@pytest.mark.parametrize("sub", ["420", "422", "444"])
@pytest.mark.parametrize("name", ["render", "noise", "ramp"])
def test_decoder_matches_reference_restatement(tool, tmp_path, name, sub):
    from kintinuous_amd import jpeg_ref
    img = _images()[name]
    h, w = img.shape[:2]
    data = jpeg_ref.encode(img, quality=90, subsampling=sub)
    got = _cxx_decode(tool, data, w, h, tmp_path)
    ref = jpeg_ref.decode(data)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} bytes differ"
    flt = jpeg_ref.decode(data, float_idct=True)
    d = np.abs(ref.astype(int) - flt.astype(int))
    assert d.max() <= 3 and (d > 1).mean() < 0.01
    if name != "noise":
        mse = np.mean((ref.astype(float) - img.astype(float)) ** 2)
        assert 10 * np.log10(255 ** 2 / mse) > 30


@pytest.mark.parametrize("kw", [dict(restart_interval=3), dict(ac_table="skewed"), dict(interleaved=False), dict(quality=35),
                                dict(restart_interval=1, subsampling="422", ac_table="skewed")])
def test_stream_layouts(tool, tmp_path, kw):
    from kintinuous_amd import jpeg_ref
    img = _images()["ramp"]
    h, w = img.shape[:2]
    data = jpeg_ref.encode(img, **kw)
    assert np.array_equal(_cxx_decode(tool, data, w, h, tmp_path), jpeg_ref.decode(data))


def test_grey_and_errors(tool, tmp_path):
    from kintinuous_amd import jpeg_ref
    img = _images()["ramp"]
    h, w = img.shape[:2]
    data = jpeg_ref.encode(img[..., 1].copy())
    got = _cxx_decode(tool, data, w, h, tmp_path)
    assert np.array_equal(got, jpeg_ref.decode(data)) and np.array_equal(got[..., 0], got[..., 2])
    good = jpeg_ref.encode(img)
    with pytest.raises(RuntimeError, match="size differs"):
        _cxx_decode(tool, good, w + 1, h, tmp_path)
    with pytest.raises(RuntimeError, match="SOI"):
        _cxx_decode(tool, b"not a jpeg at all", w, h, tmp_path)
    # truncated and bit-flipped streams must fail cleanly or decode to something -- never crash
    rng = np.random.default_rng(0)
    for k in range(40):
        bad = bytearray(good if k % 2 else good[: rng.integers(20, len(good))])
        for _ in range(3):
            bad[rng.integers(2, len(bad))] = rng.integers(0, 256)
        src, dst = tmp_path / "bad.jpg", tmp_path / "bad.bgr"
        src.write_bytes(bytes(bad))
        r = subprocess.run([tool, str(src), str(w), str(h), str(dst)], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 1), (k, r.returncode, r.stderr)


def test_klg_with_jpeg_colour_round_trip(tool, tmp_path):
    """write_klg(jpeg_quality=...) produces Logger2-style frames (zlib depth + JPEG colour) and read_klg decodes them with the same
    reference decoder the C++ reader is tested against."""
    from kintinuous_amd import jpeg_ref, klg, synth
    cam = synth.Camera.small(160, 120)
    frames = [synth.render(synth.Scene("room"), cam, *p) for p in synth.orbit_trajectory(3)]
    path = str(tmp_path / "j.klg")
    klg.write_klg(path, frames, cols=cam.cols, rows=cam.rows, compress_depth=True, jpeg_quality=90)
    back = list(klg.read_klg(path, cols=cam.cols, rows=cam.rows, reference_quirk=False))
    assert len(back) == len(frames)
    for (ts, d, rgb), (d0, rgb0) in zip(back, frames):
        assert np.array_equal(d, d0)
        assert np.array_equal(rgb, jpeg_ref.decode(jpeg_ref.encode(np.ascontiguousarray(rgb0), quality=90)))


@pytest.mark.parametrize("layout", ["raw", "zlib", "jpeg", "jpeg_flip"])
def test_cxx_log_reader(tool, tmp_path, layout):
    """The C++ RawLogReader (host/klg_tool) against the Python reader on the same log: timestamps, depth and colour bytes of every
    frame it returns (raw, zlib depth, JPEG colour, -f colour flip), and the reference's quirk that the last frame never comes out."""
    import zlib
    from kintinuous_amd import build, klg, synth
    cam = synth.Camera.small(160, 120)
    frames = [synth.render(synth.Scene("room"), cam, *p) for p in synth.orbit_trajectory(4)]
    path = str(tmp_path / "t.klg")
    klg.write_klg(path, frames, timestamps=[7 + 33333 * k for k in range(4)], cols=cam.cols, rows=cam.rows, compress_depth=layout != "raw",
                  jpeg_quality=90 if layout.startswith("jpeg") else 0)
    args = [build.KLG_TOOL, "-l", path, "-w", str(cam.cols), "-h", str(cam.rows)] + (["-f"] if layout == "jpeg_flip" else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = [l.split() for l in r.stdout.splitlines()]
    want = list(klg.read_klg(path, cols=cam.cols, rows=cam.rows))          # reference_quirk=True: 3 of the 4 frames
    assert len(lines) == len(want) == 3
    for (ts, cd, ci, comp), (wts, wd, wrgb) in zip(lines, want):
        if layout == "jpeg_flip":
            wrgb = wrgb[..., ::-1]
        assert int(ts) == wts
        assert int(cd, 16) == zlib.crc32(np.ascontiguousarray(wd, "<u2").tobytes())
        assert int(ci, 16) == zlib.crc32(np.ascontiguousarray(wrgb).tobytes())
        assert int(comp) == int(layout.startswith("jpeg"))      # the image decides isCompressed (RawLogReader.cpp:73-97)


def _pillow():
    PIL = pytest.importorskip("PIL.Image")
    return PIL


@pytest.mark.parametrize("sub", ["420", "422", "444"])
@pytest.mark.parametrize("name", ["render", "noise", "ramp"])
def test_decoder_matches_libjpeg(tool, tmp_path, name, sub):
    """The C++ decoder against libjpeg-turbo itself (through Pillow): streams written by our encoder and streams written by
    libjpeg's own encoder (different Huffman tables, sampling factors 4:2:0 / 4:2:2 / 4:4:4, qualities 60 and 92), byte for byte.
    RawLogReader keeps OpenCV's BGR order; Pillow returns RGB."""
    import io
    Image = _pillow()
    from kintinuous_amd import jpeg_ref
    img = _images()[name]
    h, w = img.shape[:2]
    streams = [jpeg_ref.encode(img, quality=90, subsampling=sub)]
    for q in (60, 92):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling={"444": 0, "422": 1, "420": 2}[sub])
        streams.append(buf.getvalue())
    for k, data in enumerate(streams):
        pil = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        got = _cxx_decode(tool, data, w, h, tmp_path, name=f"s{k}")
        assert np.array_equal(got, pil[..., ::-1]), (k, int((got != pil[..., ::-1]).sum()))
