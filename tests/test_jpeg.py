"""CPU: the .klg colour decoder (kintinuous_amd/host/JpegDecoder.h, the stand-in for cvDecodeImage in RawLogReader) against an
independent numpy restatement of libjpeg's default decode path (kintinuous_amd/jpeg_ref.py): byte-identical output for every
stream layout the encoder can produce; the integer IDCT path within 2 grey levels of a double-precision decode; decent PSNR
against the source image.  Pinned against the real library as well: Pillow (libjpeg-turbo, the decoder OpenCV's cvDecodeImage
wraps too) decodes our encoder's streams AND its own to exactly the bytes the C++ decoder produces.  The other direction too: the C++
encoder behind PlaceRecognitionInput::compress() (host/JpegEncoder.h) writes exactly the bytes libjpeg-turbo's encoder writes."""
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


def _encoder_images():
    rng = np.random.default_rng(3)
    out = dict(_images())
    out["noise48x32"] = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)           # whole MCUs
    out["noise17x9"] = rng.integers(0, 256, (9, 17, 3), dtype=np.uint8)             # dummy luma blocks right and below
    out["noise24x40"] = rng.integers(0, 256, (40, 24, 3), dtype=np.uint8)           # odd number of luma block columns
    out["pixel"] = rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)
    out["white"] = np.full((33, 31, 3), 255, np.uint8)
    return out


@pytest.mark.parametrize("name", ["render", "noise", "ramp", "noise48x32", "noise17x9", "noise24x40", "pixel", "white"])
def test_encoder_matches_libjpeg(tool, tmp_path, name):
    """PlaceRecognitionInput::compress()'s JPEG half (host/JpegEncoder.h, the stand-in for cvEncodeImage(".jpg", quality 90)) against
    libjpeg-turbo's own encoder through Pillow: the same BYTES -- markers, tables, entropy-coded data -- at the reference's quality and
    at both ends of the scale, for MCU-aligned and ragged sizes (edge replication, dummy blocks)."""
    import io
    Image = _pillow()
    bgr = np.ascontiguousarray(_encoder_images()[name])
    h, w = bgr.shape[:2]
    src = tmp_path / "in.bgr"
    src.write_bytes(bgr.tobytes())
    for q in (90, 50, 100, 5):
        dst = tmp_path / f"q{q}.jpg"
        r = subprocess.run([tool, "-e", str(src), str(w), str(h), str(q), str(dst)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(bgr[..., ::-1]), "RGB").save(buf, format="JPEG", quality=q, subsampling=2)
        assert dst.read_bytes() == buf.getvalue(), (name, q)


def test_place_recognition_input_compress_round_trip(tool, tmp_path):
    """frontend/PlaceRecognitionInput.h:72-140 through the shell's class: compress() leaves zlib depth + a quality-90 JPEG (the bytes
    libjpeg writes), decompressDepthTo returns the depth exactly, decompressImgTo what libjpeg decodes from that JPEG."""
    import io
    Image = _pillow()
    from kintinuous_amd import synth
    cam = synth.Camera.small(160, 120)
    _, rgb = synth.render(synth.Scene("room"), cam, *synth.orbit_trajectory(2)[0])
    bgr = np.ascontiguousarray(rgb)
    h, w = bgr.shape[:2]
    src, jpg, back = tmp_path / "in.bgr", tmp_path / "pr.jpg", tmp_path / "pr.bgr"
    src.write_bytes(bgr.tobytes())
    r = subprocess.run([tool, "-pr", str(src), str(w), str(h), str(jpg), str(back)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stderr)      # 3 = flags or the depth round trip wrong
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(bgr[..., ::-1]), "RGB").save(buf, format="JPEG", quality=90, subsampling=2)
    assert jpg.read_bytes() == buf.getvalue()
    pil = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    got = np.frombuffer(back.read_bytes(), np.uint8).reshape(h, w, 3)
    assert np.array_equal(got, pil[..., ::-1])
    assert len(jpg.read_bytes()) < bgr.size // 4


def _run_klg_tool(path, cols, rows, threads, extra=()):
    from kintinuous_amd import build
    args = [build.KLG_TOOL, "-l", path, "-w", str(cols), "-h", str(rows), "-dt", str(threads)] + list(extra)
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("layout", ["raw", "zlib", "jpeg"])
def test_cxx_log_reader_decode_ahead(tool, tmp_path, layout):
    """RawLogReader with decode-ahead worker threads (-dt 1 / 3 / 8 / 33) hands out exactly what the synchronous reader does: every
    frame's timestamp, depth and colour checksums and isCompressed, in order, over a log much longer than the ring of slots."""
    from kintinuous_amd import klg, synth
    cam = synth.Camera.small(160, 120)
    poses = synth.orbit_trajectory(70)
    frames = [synth.render(synth.Scene("room"), cam, *p) for p in poses[:10]]
    frames = [frames[k % 10] if k % 7 else (np.roll(frames[k % 10][0], k, axis=1), frames[k % 10][1]) for k in range(70)]
    path = str(tmp_path / "long.klg")
    klg.write_klg(path, frames, timestamps=[11 + 33333 * k for k in range(70)], cols=cam.cols, rows=cam.rows, compress_depth=layout != "raw",
                  jpeg_quality=85 if layout == "jpeg" else 0)
    rc0, out0, err0 = _run_klg_tool(path, cam.cols, cam.rows, 0)
    assert rc0 == 0 and len(out0.splitlines()) == 69, err0
    for threads in (1, 3, 8, 33):
        rc, out, err = _run_klg_tool(path, cam.cols, cam.rows, threads, ["-f"] if threads == 3 else [])
        if threads == 3:
            rc0f, out0f, _ = _run_klg_tool(path, cam.cols, cam.rows, 0, ["-f"])
            assert (rc, out) == (rc0f, out0f)
        else:
            assert (rc, out) == (rc0, out0), (threads, err)


@pytest.mark.parametrize("threads", [0, 1, 3, 8])
def test_cxx_log_reader_frame_lifetime(tool, tmp_path, threads):
    """A frame handed out by grabNext stays untouched for three further grabNext calls (the tracker keeps the current frame and two
    earlier ones), with the synchronous reader and with every number of decode-ahead workers: klg_tool -hold re-checks the held
    buffers after every read (the consumer is slower than the decoders here, so they run as far ahead as the ring lets them)."""
    from kintinuous_amd import klg, synth
    cam = synth.Camera.small(160, 120)
    base = [synth.render(synth.Scene("room"), cam, *p) for p in synth.orbit_trajectory(6)]
    frames = [(np.roll(base[k % 6][0], k, axis=1), np.roll(base[k % 6][1], k, axis=0)) for k in range(60)]
    path = str(tmp_path / "hold.klg")
    klg.write_klg(path, frames, cols=cam.cols, rows=cam.rows, compress_depth=True)
    rc, out, err = _run_klg_tool(path, cam.cols, cam.rows, threads, ["-hold"])
    assert rc == 0 and len(out.splitlines()) == 59, err


@pytest.mark.parametrize("damage", ["truncated_payload", "truncated_header", "bad_zlib", "bad_jpeg", "bad_sizes", "short_count"])
def test_cxx_log_reader_decode_ahead_on_damaged_logs(tool, tmp_path, damage):
    """A damaged log stops the decode-ahead reader at the same frame, with the same frames delivered before it, the same exit code and
    the same message as the synchronous reader (a truncated file ends the log quietly, a corrupt frame is an error)."""
    import struct
    from kintinuous_amd import klg, synth
    cam = synth.Camera.small(160, 120)
    fr = [synth.render(synth.Scene("room"), cam, *p) for p in synth.orbit_trajectory(4)]
    frames = [fr[k % 4] for k in range(24)]
    path = str(tmp_path / "d.klg")
    klg.write_klg(path, frames, timestamps=[5 + 1000 * k for k in range(24)], cols=cam.cols, rows=cam.rows, compress_depth=True, jpeg_quality=85)
    data = bytearray(open(path, "rb").read())
    # walk the records to find frame 13
    off, starts = 4, []
    for k in range(24):
        starts.append(off)
        ds, is_ = struct.unpack_from("<ii", data, off + 8)
        off += 16 + ds + is_
    s13 = starts[13]
    ds13, is13 = struct.unpack_from("<ii", data, s13 + 8)
    if damage == "truncated_payload":
        data = data[:s13 + 16 + ds13 // 2]
    elif damage == "truncated_header":
        data = data[:s13 + 10]
    elif damage == "bad_zlib":
        for i in range(s13 + 16 + 20, s13 + 16 + 60):
            data[i] ^= 0x5A
    elif damage == "bad_jpeg":
        j = s13 + 16 + ds13
        data[j:j + 2] = b"\x00\x00"      # no SOI marker
    elif damage == "bad_sizes":
        struct.pack_into("<ii", data, s13 + 8, -5, 1 << 30)
    elif damage == "short_count":
        struct.pack_into("<i", data, 0, 9)   # the header announces fewer frames than the file holds
    open(path, "wb").write(bytes(data))
    rc0, out0, err0 = _run_klg_tool(path, cam.cols, cam.rows, 0)
    want_frames = {"short_count": 8}.get(damage, 13)
    assert len(out0.splitlines()) == want_frames, (damage, rc0, err0)
    assert (rc0 == 0) == (damage in ("truncated_payload", "truncated_header", "short_count"))
    for threads in (1, 4, 16):
        rc, out, err = _run_klg_tool(path, cam.cols, cam.rows, threads)
        assert (rc, out, err) == (rc0, out0, err0), (damage, threads, rc, err)
