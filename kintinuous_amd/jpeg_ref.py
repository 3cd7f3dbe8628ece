"""Baseline JPEG in numpy: an encoder (to make Logger2-style .klg colour payloads and test streams) and a reference decoder that
restates libjpeg's default decode path (ISLOW integer IDCT, "fancy" triangle chroma upsampling, fixed-point YCbCr -> RGB) with
vectorised integer arithmetic.  Test infrastructure for kintinuous_amd/host/JpegDecoder.h -- written independently of it (different
structure: whole-plane numpy ops, table-driven Huffman) so that the two can be compared bit for bit.  tests/test_jpeg.py also compares both with libjpeg-turbo
(through Pillow).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                   62, 63])

# ITU T.81 Annex K tables
Q_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87,
                   80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92,
                   95, 98, 112, 100, 103, 99])
Q_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99,
                     99, 99, 99] + [99] * 32)
DC_LUMA_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
DC_CHROMA_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
DC_VALS = list(range(12))


def _flat_ac_table() -> Tuple[List[int], List[int]]:
    """A valid (if wasteful) AC table: the 162 run/size symbols, all with 8-bit codes."""
    vals = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    bits = [0] * 16
    bits[7] = len(vals)
    return bits, vals


def _skewed_ac_table() -> Tuple[List[int], List[int]]:
    """Another valid AC table with code lengths 2..16, to exercise every length of the decoder's canonical tables."""
    vals = [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x12, 0x21, 0xF0] + [v for v in _flat_ac_table()[1] if v not in (0, 1, 2, 3, 0x11, 4, 0x12, 0x21, 0xF0)]
    bits = [0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0]      # one code each for lengths 2..15
    bits[15] = len(vals) - sum(bits)                          # everything else at 16 bits
    return bits, vals


def _codes(bits: List[int], vals: List[int]) -> Dict[int, Tuple[int, int]]:
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class _BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, code: int, length: int) -> None:
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.buf.append(b)
            if b == 0xFF:
                self.buf.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self) -> None:
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _dct_matrix() -> np.ndarray:
    k = np.arange(8)
    m = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * 0.5
    m[0] *= 1 / np.sqrt(2)
    return m


_D = _dct_matrix()


def _blocks(plane: np.ndarray) -> np.ndarray:
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)


def _magnitude(v: int) -> Tuple[int, int]:
    a = abs(v)
    size = a.bit_length()
    return size, (v if v >= 0 else v + (1 << size) - 1)


def encode(bgr: np.ndarray, quality: int = 90, subsampling: str = "420", restart_interval: int = 0, ac_table: str = "flat",
           interleaved: bool = True) -> bytes:
    """bgr uint8 [H, W, 3] (or [H, W] for a grey stream) -> baseline JFIF bytes.  subsampling: "444" | "422" | "420"."""
    grey = bgr.ndim == 2
    H, W = bgr.shape[:2]
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    qtabs = [np.clip((q * scale + 50) // 100, 1, 255) for q in (Q_LUMA, Q_CHROMA)]
    if grey:
        planes, samp = [bgr.astype(np.float64)], [(1, 1)]
    else:
        b, g, r = [bgr[..., i].astype(np.float64) for i in range(3)]
        y = 0.299 * r + 0.587 * g + 0.114 * b
        cb = -0.168735892 * r - 0.331264108 * g + 0.5 * b + 128
        cr = 0.5 * r - 0.418687589 * g - 0.081312411 * b + 128
        hs, vs = {"444": (1, 1), "422": (2, 1), "420": (2, 2)}[subsampling]
        planes, samp = [y, cb, cr], [(hs, vs), (1, 1), (1, 1)]
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    coefs = []
    for p, (h, v) in zip(planes, samp):
        fh, fv = hmax // h, vmax // v
        ph, pw = -(-H // fv) * fv, -(-W // fh) * fh
        p = np.pad(p, ((0, ph - H), (0, pw - W)), mode="edge")
        p = p.reshape(ph // fv, fv, pw // fh, fh).mean(axis=(1, 3))                      # box down-sampling
        p = np.pad(p, ((0, mcuy * v * 8 - p.shape[0]), (0, mcux * h * 8 - p.shape[1])), mode="edge")
        blk = _blocks(p - 128.0)
        c = np.einsum("ij,abjk,lk->abil", _D, blk, _D)
        q = qtabs[0 if len(coefs) == 0 else 1].reshape(8, 8)
        coefs.append(np.rint(c / q).astype(np.int64).reshape(blk.shape[0], blk.shape[1], 64)[..., ZIGZAG])

    ac_bits, ac_vals = _flat_ac_table() if ac_table == "flat" else _skewed_ac_table()
    dc_codes = [_codes(DC_LUMA_BITS, DC_VALS), _codes(DC_CHROMA_BITS, DC_VALS)]
    ac_codes = _codes(ac_bits, ac_vals)

    out = bytearray(b"\xFF\xD8")
    out += b"\xFF\xE0" + (16).to_bytes(2, "big") + b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    for i, q in enumerate(qtabs[:1 if grey else 2]):
        out += b"\xFF\xDB" + (67).to_bytes(2, "big") + bytes([i]) + bytes(int(v) for v in q[ZIGZAG])
    out += b"\xFF\xC0" + (8 + 3 * len(planes)).to_bytes(2, "big") + b"\x08" + H.to_bytes(2, "big") + W.to_bytes(2, "big") + bytes([len(planes)])
    for i, (h, v) in enumerate(samp):
        out += bytes([i + 1, (h << 4) | v, 0 if i == 0 else 1])
    for cls, idx, bits, vals in ((0, 0, DC_LUMA_BITS, DC_VALS), (0, 1, DC_CHROMA_BITS, DC_VALS), (1, 0, ac_bits, ac_vals)):
        out += b"\xFF\xC4" + (19 + len(vals)).to_bytes(2, "big") + bytes([(cls << 4) | idx]) + bytes(bits) + bytes(vals)
    if restart_interval:
        out += b"\xFF\xDD\x00\x04" + restart_interval.to_bytes(2, "big")

    def put_block(bw, zz, pred, dcc):
        size, bitsv = _magnitude(int(zz[0]) - pred)
        bw.put(*dcc[size])
        if size:
            bw.put(bitsv, size)
        run = 0
        last = int(np.max(np.nonzero(zz)[0])) if np.any(zz[1:]) else 0
        for k in range(1, last + 1):
            v = int(zz[k])
            if v == 0:
                run += 1
                continue
            while run > 15:
                bw.put(*ac_codes[0xF0])
                run -= 16
            size, bitsv = _magnitude(v)
            assert size <= 10
            bw.put(*ac_codes[(run << 4) | size])
            bw.put(bitsv, size)
            run = 0
        if last < 63:
            bw.put(*ac_codes[0x00])
        return int(zz[0])

    def scan(comp_ids):
        nonlocal out
        out += b"\xFF\xDA" + (6 + 2 * len(comp_ids)).to_bytes(2, "big") + bytes([len(comp_ids)])
        for c in comp_ids:
            out += bytes([c + 1, ((0 if c == 0 else 1) << 4) | 0])
        out += b"\x00\x3F\x00"
        bw = _BitWriter()
        preds = {c: 0 for c in comp_ids}
        if len(comp_ids) > 1:
            units = [(my, mx) for my in range(mcuy) for mx in range(mcux)]
        else:
            h, v = samp[comp_ids[0]]
            cw, ch = -(-W * h // hmax), -(-H * v // vmax)
            units = [(by, bx) for by in range(-(-ch // 8)) for bx in range(-(-cw // 8))]
        rst = 0
        for n, (uy, ux) in enumerate(units):
            if restart_interval and n and n % restart_interval == 0:
                bw.flush()
                bw.buf += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                preds = {c: 0 for c in comp_ids}
            for c in comp_ids:
                h, v = samp[c] if len(comp_ids) > 1 else (1, 1)
                for by in range(v):
                    for bx in range(h):
                        preds[c] = put_block(bw, coefs[c][uy * v + by, ux * h + bx], preds[c], dc_codes[0 if c == 0 else 1])
        bw.flush()
        out += bw.buf

    if interleaved or grey:
        scan(list(range(len(planes))))
    else:
        for c in range(len(planes)):
            scan([c])
    out += b"\xFF\xD9"
    return bytes(out)


# ---- reference decoder (libjpeg defaults, vectorised integer arithmetic) -----------------------------------------------------------
def _parse(data: bytes):
    assert data[:2] == b"\xFF\xD8"
    p, q, huff, frame, dri, scans = 2, {}, {}, None, 0, []
    while p < len(data):
        assert data[p] == 0xFF, hex(data[p])
        m = data[p + 1]
        p += 2
        if m == 0xD9:
            break
        n = int.from_bytes(data[p:p + 2], "big")
        s = data[p + 2:p + n]
        if m == 0xDB:
            i = 0
            while i < len(s):
                pq, tq = s[i] >> 4, s[i] & 15
                assert pq == 0
                t = np.zeros(64, np.int64)
                t[ZIGZAG] = np.frombuffer(s[i + 1:i + 65], np.uint8)
                q[tq] = t
                i += 65
        elif m == 0xC4:
            i = 0
            while i < len(s):
                tc, th = s[i] >> 4, s[i] & 15
                bits = list(s[i + 1:i + 17])
                vals = list(s[i + 17:i + 17 + sum(bits)])
                huff[(tc, th)] = {(code, length): sym for sym, (code, length) in _codes(bits, vals).items()}
                i += 17 + sum(bits)
        elif m in (0xC0, 0xC1):
            H, W, nc = int.from_bytes(s[1:3], "big"), int.from_bytes(s[3:5], "big"), s[5]
            frame = (H, W, [(s[6 + 3 * c], s[7 + 3 * c] >> 4, s[7 + 3 * c] & 15, s[8 + 3 * c]) for c in range(nc)])
        elif m == 0xDD:
            dri = int.from_bytes(s[:2], "big")
        elif m == 0xDA:
            ns = s[0]
            sel = [(s[1 + 2 * i], s[2 + 2 * i] >> 4, s[2 + 2 * i] & 15) for i in range(ns)]
            e = p + n
            while not (data[e] == 0xFF and data[e + 1] != 0 and not 0xD0 <= data[e + 1] <= 0xD7):
                e += 1
            scans.append((sel, data[p + n:e], dict(huff), dri))
            p = e
            continue
        p += n
    return q, frame, scans


class _Bits:
    def __init__(self, seg: bytes):
        # unstuff and split at restart markers
        self.chunks, cur, i = [], bytearray(), 0
        while i < len(seg):
            if seg[i] == 0xFF:
                if seg[i + 1] == 0:
                    cur.append(0xFF)
                elif 0xD0 <= seg[i + 1] <= 0xD7:
                    self.chunks.append(bytes(cur))
                    cur = bytearray()
                i += 2
                continue
            cur.append(seg[i])
            i += 1
        self.chunks.append(bytes(cur))
        self.ci = -1
        self.next_chunk()

    def next_chunk(self):
        self.ci += 1
        self.bits = np.unpackbits(np.frombuffer(self.chunks[self.ci], np.uint8)).tolist() + [0] * 64
        self.pos = 0

    def get(self, n: int) -> int:
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bits[self.pos]
            self.pos += 1
        return v

    def symbol(self, table) -> int:
        code = 0
        for length in range(1, 17):
            code = (code << 1) | self.bits[self.pos]
            self.pos += 1
            if (code, length) in table:
                return table[(code, length)]
        raise ValueError("bad Huffman code")


def _extend(v: int, t: int) -> int:
    return 0 if t == 0 else (v - (1 << t) + 1 if v < (1 << (t - 1)) else v)


def _idct_islow(c: np.ndarray) -> np.ndarray:
    """c: int64 [..., 8, 8] dequantised coefficients (row, col) -> samples uint8 [..., 8, 8] (jidctint.c, both passes vectorised)."""
    CB, P1 = 13, 2
    F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)

    def descale(x, n):
        return (x + (1 << (n - 1))) >> n

    def pass_(x, shift, first):
        # x[..., k, :]: frequency k along the axis being transformed
        z2, z3 = x[..., 2, :], x[..., 6, :]
        z1 = (z2 + z3) * F["c"]
        tmp2 = z1 + z3 * (-F["h"])
        tmp3 = z1 + z2 * F["d"]
        tmp0 = (x[..., 0, :] + x[..., 4, :]) << CB
        tmp1 = (x[..., 0, :] - x[..., 4, :]) << CB
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        t0, t1, t2, t3 = x[..., 7, :], x[..., 5, :], x[..., 3, :], x[..., 1, :]
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F["f"]
        t0, t1, t2, t3 = t0 * F["a"], t1 * F["j"], t2 * F["l"], t3 * F["g"]
        z1, z2, z3, z4 = z1 * -F["e"], z2 * -F["k"], z3 * -F["i"] + z5, z4 * -F["b"] + z5
        t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
        out = np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3], axis=-2)
        out = descale(out, shift)
        # the zero-AC shortcut of each pass uses a different (exact) expression: dc << 2 in pass 1, descale(dc, 5) in pass 2
        zero_ac = np.all(x[..., 1:, :] == 0, axis=-2, keepdims=True)
        dc = x[..., 0:1, :]
        short = (dc << P1) if first else descale(dc, P1 + 3)
        return np.where(zero_ac, np.broadcast_to(short, out.shape), out)

    ws = pass_(c, CB - P1, True)                                    # columns: frequency index = row
    rows = pass_(np.swapaxes(ws, -1, -2), CB + P1 + 3, False)       # rows: frequency index = column
    s = np.swapaxes(rows, -1, -2)
    x = ((s & 1023) ^ 512) - 512                                    # the & RANGE_MASK wrap of the table index, sign restored
    return np.clip(x + 128, 0, 255).astype(np.uint8)


def _fancy_h2(p: np.ndarray) -> np.ndarray:
    p = p.astype(np.int64)
    w = p.shape[1]
    out = np.empty((p.shape[0], 2 * w), np.int64)
    out[:, 0] = p[:, 0]
    out[:, 2 * w - 1] = p[:, -1]
    out[:, 2:2 * w:2] = (3 * p[:, 1:] + p[:, :-1] + 1) >> 2
    out[:, 1:2 * w - 1:2] = (3 * p[:, :-1] + p[:, 1:] + 2) >> 2
    return out


def _fancy_h2v2(p: np.ndarray) -> np.ndarray:
    p = p.astype(np.int64)
    h, w = p.shape
    up = np.vstack([p[:1], p[:-1]])
    dn = np.vstack([p[1:], p[-1:]])
    out = np.empty((2 * h, 2 * w), np.int64)
    for r, far in ((0, up), (1, dn)):
        cs = 3 * p + far                                             # column sums
        o = np.empty((h, 2 * w), np.int64)
        o[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        o[:, 2 * w - 1] = (cs[:, -1] * 4 + 7) >> 4
        o[:, 2:2 * w:2] = (3 * cs[:, 1:] + cs[:, :-1] + 8) >> 4
        o[:, 1:2 * w - 1:2] = (3 * cs[:, :-1] + cs[:, 1:] + 7) >> 4
        out[r::2] = o
    return out


def decode(data: bytes, float_idct: bool = False) -> np.ndarray:
    """-> uint8 [H, W, 3] B G R (grey streams replicated), as cvDecodeImage returns it.  float_idct = a textbook double-precision
    IDCT instead of jidctint.c (sanity reference: within a grey level or two of the integer path)."""
    q, (H, W, comps), scans = _parse(data)
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    coef = {cid: np.zeros((mcuy * v, mcux * h, 64), np.int64) for cid, h, v, _ in comps}
    geom = {cid: (h, v, tq) for cid, h, v, tq in comps}
    for sel, seg, huff, dri in scans:
        br = _Bits(seg)
        pred = {cid: 0 for cid, _, _ in sel}
        if len(sel) > 1:
            units = [(my, mx) for my in range(mcuy) for mx in range(mcux)]
        else:
            h, v, _ = geom[sel[0][0]]
            units = [(by, bx) for by in range(-(-(-(-H * v // vmax)) // 8)) for bx in range(-(-(-(-W * h // hmax)) // 8))]
        for n, (uy, ux) in enumerate(units):
            if dri and n and n % dri == 0:
                br.next_chunk()
                pred = {cid: 0 for cid in pred}
            for cid, td, ta in sel:
                h, v, _ = geom[cid] if len(sel) > 1 else (1, 1, 0)
                for by in range(v):
                    for bx in range(h):
                        zz = np.zeros(64, np.int64)
                        t = br.symbol(huff[(0, td)])
                        pred[cid] += _extend(br.get(t), t)
                        zz[0] = pred[cid]
                        k = 1
                        while k < 64:
                            rs = br.symbol(huff[(1, ta)])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                break
                            k += r
                            zz[k] = _extend(br.get(s), s)
                            k += 1
                        nat = np.zeros(64, np.int64)
                        nat[ZIGZAG] = zz
                        coef[cid][uy * v + by, ux * h + bx] = nat
    planes = []
    for cid, h, v, tq in comps:
        c = coef[cid] * q[tq]
        c = c.reshape(c.shape[0], c.shape[1], 8, 8)
        if float_idct:
            s = np.einsum("ji,abjk,kl->abil", _D, c.astype(np.float64), _D)
            blk = np.clip(np.rint(s + 128), 0, 255).astype(np.uint8)
        else:
            blk = _idct_islow(c)
        plane = blk.transpose(0, 2, 1, 3).reshape(c.shape[0] * 8, c.shape[1] * 8)
        cw, ch = -(-W * h // hmax), -(-H * v // vmax)
        plane = plane[:ch, :cw]
        hr, vr = hmax // h, vmax // v
        if (hr, vr) == (1, 1):
            full = plane.astype(np.int64)
        elif (hr, vr) == (2, 1) and cw > 2:
            full = _fancy_h2(plane)
        elif (hr, vr) == (2, 2) and cw > 2:
            full = _fancy_h2v2(plane)
        else:
            full = np.repeat(np.repeat(plane.astype(np.int64), vr, axis=0), hr, axis=1)
        planes.append(full[:H, :W])
    if len(planes) == 1:
        g = planes[0].astype(np.uint8)
        return np.stack([g, g, g], axis=-1)
    y, cb, cr = planes
    x_cb, x_cr = cb - 128, cr - 128
    r = y + ((91881 * x_cr + 32768) >> 16)
    g = y + ((-22554 * x_cb + 32768 - 46802 * x_cr) >> 16)
    b = y + ((116130 * x_cb + 32768) >> 16)
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)
