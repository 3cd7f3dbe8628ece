// JpegDecoder.h -- baseline JPEG -> BGR24, for the colour payload of .klg logs.
//
// The reference decodes it with cvDecodeImage (utils/RawLogReader.cpp:85), i.e. OpenCV 2.4 -> the system libjpeg of Ubuntu 14.04 /
// 15.04 (README.md:14-31), which is libjpeg-turbo with the library defaults: dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE,
// out_color_space = JCS_RGB, then OpenCV's RGB -> BGR swap.  Neither library is in this image, so this is a restatement of those
// published algorithms (IJG libjpeg 6b / libjpeg-turbo: jidctint.c "slow-but-accurate integer IDCT", jdsample.c "fancy" triangle
// upsampling for 2h1v and 2h2v, jdcolor.c fixed-point YCbCr -> RGB), written to give the same bytes.  Pinned: tests/test_jpeg.py
// compares it byte for byte with libjpeg-turbo itself (through Pillow, which bundles it) on 4:4:4 / 4:2:2 / 4:2:0 / grey streams from
// two encoders, and with an independent numpy restatement of the same integer algorithms.
//
// Supported: SOF0 / SOF1 (8-bit sequential Huffman), 1 or 3 components, sampling factors 1 or 2, interleaved or per-component scans,
// restart intervals, 8- and 16-bit quantisation tables.  Progressive, arithmetic, lossless, CMYK: rejected with a message.
// Input is untrusted (a log file): every read is bounds-checked.
#pragma once

#include <stdint.h>
#include <cstring>
#include <string>
#include <vector>

namespace kt {
namespace jpeg {

namespace detail {

static const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                          41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                          30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present;
    // canonical decoding (ITU T.81 F.2.2.3): per code length the smallest code, the largest code and the index of its first symbol
    int mincode[17], maxcode[18], valptr[17];
    unsigned char vals[256];
    int nvals;
    // look-ahead for the short codes: entry = code length << 8 | symbol for every kLookBits-bit window that starts with that code, 0 = longer
    static const int kLookBits = 9;
    unsigned short look[1 << kLookBits];
    Huffman() : present(false), nvals(0) { std::memset(look, 0, sizeof(look)); }
};

struct Component {
    int id, h, v, tq, td, ta;
    int blocksW, blocksH;          // allocated plane, in blocks (whole MCUs)
    int width, height;             // downsampled_width / downsampled_height: ceil(image * h / hmax)
    int pred;
    std::vector<unsigned char> plane;   // blocksW * 8 columns
    bool decoded;
};

// sample_range_limit of jdmaster.c (prepare_range_limit_table), indexed through (x & 1023) with x = sample - 128
struct RangeLimit {
    unsigned char t[1024];
    RangeLimit()
    {
        for (int i = 0; i < 1024; ++i) {
            const int x = i < 512 ? i : i - 1024;   // the value before the mask
            const int s = x + 128;
            t[i] = (unsigned char)(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
        // the real table holds 255 only up to x = 511 - 128 and zeros beyond; after-IDCT values of valid streams stay far inside
    }
};

// Entropy-coded segment reader: a 64-bit window topped up a byte at a time (0xFF00 un-stuffed; a marker or the end of the data stops
// the consumption and zero bits follow -- libjpeg feeds zeros there too: a corrupt tail decodes to grey, not a crash).
class BitReader {
  public:
    BitReader(const unsigned char* d, size_t n) : d(d), n(n), pos(0), acc(0), cnt(0), marker(0) {}
    size_t pos_bytes() const { return pos; }   // never past a marker: the window stops in front of it
    void seek(size_t p) { pos = p; acc = 0; cnt = 0; marker = 0; }
    // the next k bits (k <= 16) without consuming them
    unsigned int peek(int k)
    {
        if (cnt < k) refill();
        return (unsigned int)(acc >> (cnt - k)) & ((1u << k) - 1u);
    }
    void skip(int k) { cnt -= k; }
    int bits(int k)
    {
        if (k == 0) return 0;
        const int v = (int)peek(k);
        cnt -= k;
        return v;
    }
    // after an MCU row / restart interval: drop the partial byte and consume an RSTn marker if one follows
    bool restart()
    {
        cnt = 0; acc = 0;
        // skip fill bytes
        while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) {
            if (d[pos] == 0xFF && d[pos + 1] != 0x00 && d[pos + 1] != 0xFF) return false;   // some other marker
            ++pos;
        }
        if (pos + 1 >= n) return false;
        pos += 2;
        marker = 0;
        return true;
    }
    bool hit_marker() const { return marker != 0; }

  private:
    void refill()
    {
        while (cnt <= 56) {
            unsigned int b = 0;
            if (!marker && pos < n) {
                b = d[pos++];
                if (b == 0xFF) {
                    if (pos < n && d[pos] == 0x00) ++pos;          // stuffed zero
                    else { marker = (pos < n) ? d[pos] : 0xD9; --pos; b = 0; }   // a marker: stop consuming, feed zeros
                }
            }
            acc = (acc << 8) | b;
            cnt += 8;
        }
    }
    const unsigned char* d;
    size_t n, pos;
    unsigned long long acc;   // the low cnt bits are the unread ones, oldest on top
    int cnt;
    int marker;
};

// One symbol: codes of up to kLookBits bits come out of Huffman::look (length << 8 | symbol, indexed by the next kLookBits bits), longer
// ones out of the canonical search of T.81 F.2.2.3 continued from there.
inline int decode_symbol(BitReader& br, const Huffman& h)
{
    const unsigned int w = br.peek(16);
    const unsigned int e = h.look[w >> (16 - Huffman::kLookBits)];
    if (e) {
        br.skip((int)(e >> 8));
        return (int)(e & 0xffu);
    }
    for (int len = Huffman::kLookBits + 1; len <= 16; ++len) {
        const int code = (int)(w >> (16 - len));
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
            const int idx = h.valptr[len] + code - h.mincode[len];
            br.skip(len);
            return idx < h.nvals ? h.vals[idx] : -1;
        }
    }
    return -1;
}

inline int extend(int v, int t) { return (t == 0) ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }   // T.81 F.2.2.1

// jpeg_idct_islow (jidctint.c), dequantisation included; out = 8 rows of 8 samples at `stride`
inline void idct_islow(const int* coef /* natural order */, const unsigned short* q, unsigned char* out, int stride, const RangeLimit& rl)
{
    const int CONST_BITS = 13, PASS1_BITS = 2;
    const int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
              F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
              F_2_562915447 = 20995, F_3_072711026 = 25172;
    // INT32 in libjpeg; 64-bit here so that a corrupt stream (huge coefficients) cannot overflow -- identical results otherwise
    typedef long long acc;
#define KT_DESCALE(x, n) (((x) + ((acc)1 << ((n)-1))) >> (n))
    acc ws[64];
    for (int c = 0; c < 8; ++c) {
        const int* in = coef + c;
        const unsigned short* qq = q + c;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
            const acc dc = ((acc)in[0] * (acc)qq[0]) * (1 << PASS1_BITS);
            for (int r = 0; r < 8; ++r) ws[r * 8 + c] = dc;
            continue;
        }
        acc z2 = (acc)in[16] * (acc)qq[16], z3 = (acc)in[48] * (acc)qq[48];
        acc z1 = (z2 + z3) * F_0_541196100;
        acc tmp2 = z1 + z3 * (-F_1_847759065);
        acc tmp3 = z1 + z2 * F_0_765366865;
        z2 = (acc)in[0] * (acc)qq[0]; z3 = (acc)in[32] * (acc)qq[32];
        acc tmp0 = (z2 + z3) * (1 << CONST_BITS);
        acc tmp1 = (z2 - z3) * (1 << CONST_BITS);
        const acc tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (acc)in[56] * (acc)qq[56]; tmp1 = (acc)in[40] * (acc)qq[40]; tmp2 = (acc)in[24] * (acc)qq[24]; tmp3 = (acc)in[8] * (acc)qq[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        acc z4 = tmp1 + tmp3;
        const acc z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[0 * 8 + c] = KT_DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
        ws[7 * 8 + c] = KT_DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        ws[1 * 8 + c] = KT_DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
        ws[6 * 8 + c] = KT_DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        ws[2 * 8 + c] = KT_DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
        ws[5 * 8 + c] = KT_DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        ws[3 * 8 + c] = KT_DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
        ws[4 * 8 + c] = KT_DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) {
        const acc* w = ws + r * 8;
        unsigned char* o = out + (size_t)r * stride;
        if (w[1] == 0 && w[2] == 0 && w[3] == 0 && w[4] == 0 && w[5] == 0 && w[6] == 0 && w[7] == 0) {
            const unsigned char dc = rl.t[KT_DESCALE(w[0], PASS1_BITS + 3) & 1023];
            for (int c = 0; c < 8; ++c) o[c] = dc;
            continue;
        }
        acc z2 = w[2], z3 = w[6];
        acc z1 = (z2 + z3) * F_0_541196100;
        acc tmp2 = z1 + z3 * (-F_1_847759065);
        acc tmp3 = z1 + z2 * F_0_765366865;
        acc tmp0 = (w[0] + w[4]) * (1 << CONST_BITS);
        acc tmp1 = (w[0] - w[4]) * (1 << CONST_BITS);
        const acc tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        acc z4 = tmp1 + tmp3;
        const acc z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int S = CONST_BITS + PASS1_BITS + 3;
        o[0] = rl.t[KT_DESCALE(tmp10 + tmp3, S) & 1023];
        o[7] = rl.t[KT_DESCALE(tmp10 - tmp3, S) & 1023];
        o[1] = rl.t[KT_DESCALE(tmp11 + tmp2, S) & 1023];
        o[6] = rl.t[KT_DESCALE(tmp11 - tmp2, S) & 1023];
        o[2] = rl.t[KT_DESCALE(tmp12 + tmp1, S) & 1023];
        o[5] = rl.t[KT_DESCALE(tmp12 - tmp1, S) & 1023];
        o[3] = rl.t[KT_DESCALE(tmp13 + tmp0, S) & 1023];
        o[4] = rl.t[KT_DESCALE(tmp13 - tmp0, S) & 1023];
    }
#undef KT_DESCALE
}

// one row of h2v1_fancy_upsample (jdsample.c): 3/4 nearer + 1/4 further, rounding alternately down / up
inline void fancy_h2_row(const unsigned char* in, int w, unsigned char* out)
{
    int v = in[0];
    out[0] = (unsigned char)v;
    out[1] = (unsigned char)((v * 3 + in[1] + 2) >> 2);
    for (int c = 1; c < w - 1; ++c) {
        v = in[c] * 3;
        out[2 * c] = (unsigned char)((v + in[c - 1] + 1) >> 2);
        out[2 * c + 1] = (unsigned char)((v + in[c + 1] + 2) >> 2);
    }
    v = in[w - 1];
    out[2 * (w - 1)] = (unsigned char)((v * 3 + in[w - 2] + 1) >> 2);
    out[2 * (w - 1) + 1] = (unsigned char)v;
}

// one output row of h2v2_fancy_upsample: near = the input row it belongs to, far = the row above (upper output row) or below
inline void fancy_h2v2_row(const unsigned char* near, const unsigned char* far, int w, unsigned char* out)
{
    int thiscol = near[0] * 3 + far[0];
    int nextcol = near[1] * 3 + far[1];
    out[0] = (unsigned char)((thiscol * 4 + 8) >> 4);
    out[1] = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
    int lastcol = thiscol;
    thiscol = nextcol;
    for (int c = 1; c < w - 1; ++c) {
        nextcol = near[c + 1] * 3 + far[c + 1];
        out[2 * c] = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
        out[2 * c + 1] = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol;
        thiscol = nextcol;
    }
    out[2 * (w - 1)] = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
    out[2 * (w - 1) + 1] = (unsigned char)((thiscol * 4 + 7) >> 4);
}

}  // namespace detail

// Decodes `data` into `bgr` (width * height * 3 bytes, rows top-down, B G R).  The image must have exactly the expected size.
inline bool decodeBGR(const unsigned char* data, size_t size, int width, int height, unsigned char* bgr, std::string* err)
{
    using namespace detail;
#define KT_FAIL(msg) do { if (err) *err = (msg); return false; } while (0)
    static const RangeLimit rl;
    unsigned short qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huffman hdc[4], hac[4];
    std::vector<Component> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0;
    bool have_sof = false;

    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) KT_FAIL("not a JPEG stream (no SOI)");
    size_t p = 2;
    bool done = false;
    while (!done) {
        // next marker
        while (p < size && data[p] != 0xFF) ++p;
        while (p < size && data[p] == 0xFF) ++p;
        if (p >= size) break;
        const int m = data[p++];
        if (m == 0xD9) break;                                     // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;      // TEM, stray RSTn: no payload
        if (p + 2 > size) KT_FAIL("truncated marker segment");
        const size_t len = ((size_t)data[p] << 8) | data[p + 1];
        if (len < 2 || p + len > size) KT_FAIL("bad marker segment length");
        const unsigned char* s = data + p + 2;
        const size_t n = len - 2;
        switch (m) {
        case 0xDB: {  // DQT
            size_t i = 0;
            while (i < n) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || pq > 1) KT_FAIL("bad DQT");
                if (i + (pq ? 128 : 64) > n) KT_FAIL("truncated DQT");
                for (int k = 0; k < 64; ++k) {
                    const unsigned short v = pq ? (unsigned short)((s[i] << 8) | s[i + 1]) : s[i];
                    i += pq ? 2 : 1;
                    qt[tq][kZigzag[k]] = v;
                }
                qt_present[tq] = true;
            }
            break;
        }
        case 0xC4: {  // DHT
            size_t i = 0;
            while (i < n) {
                if (i + 17 > n) KT_FAIL("truncated DHT");
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (tc > 1 || th > 3) KT_FAIL("bad DHT");
                int bitsn[17], total = 0;
                for (int k = 1; k <= 16; ++k) { bitsn[k] = s[i + k]; total += bitsn[k]; }
                i += 17;
                if (total > 256 || i + total > n) KT_FAIL("bad DHT symbol count");
                Huffman& h = tc ? hac[th] : hdc[th];
                h.nvals = total;
                std::memcpy(h.vals, s + i, total);
                i += total;
                int code = 0, k = 0;
                for (int l = 1; l <= 16; ++l) {
                    h.valptr[l] = k;
                    h.mincode[l] = code;
                    code += bitsn[l];
                    k += bitsn[l];
                    h.maxcode[l] = bitsn[l] ? code - 1 : -1;
                    if (code > (1 << l)) KT_FAIL("DHT code lengths overflow");
                    code <<= 1;
                }
                h.maxcode[17] = -1;
                std::memset(h.look, 0, sizeof(h.look));
                for (int l = 1; l <= Huffman::kLookBits; ++l)
                    for (int c = h.mincode[l]; c <= h.maxcode[l]; ++c) {   // (empty lengths have maxcode -1)
                        const int idx = h.valptr[l] + c - h.mincode[l];
                        if (idx >= h.nvals) continue;
                        const int first = c << (Huffman::kLookBits - l);
                        for (int f = 0; f < (1 << (Huffman::kLookBits - l)); ++f) h.look[first + f] = (unsigned short)((l << 8) | h.vals[idx]);
                    }
                h.present = true;
            }
            break;
        }
        case 0xC0: case 0xC1: {  // SOF0 / SOF1
            if (have_sof) KT_FAIL("two frame headers");
            if (n < 6) KT_FAIL("truncated SOF");
            if (s[0] != 8) KT_FAIL("only 8-bit JPEG is supported");
            H = (s[1] << 8) | s[2];
            W = (s[3] << 8) | s[4];
            const int nc = s[5];
            if (nc != 1 && nc != 3) KT_FAIL("only grey and YCbCr JPEG are supported");
            if (n < (size_t)(6 + 3 * nc)) KT_FAIL("truncated SOF");
            if (W != width || H != height) KT_FAIL("JPEG size differs from the log resolution");
            comps.resize(nc);
            for (int c = 0; c < nc; ++c) {
                Component& k = comps[c];
                k.id = s[6 + 3 * c];
                k.h = s[7 + 3 * c] >> 4;
                k.v = s[7 + 3 * c] & 15;
                k.tq = s[8 + 3 * c];
                if (k.h < 1 || k.h > 2 || k.v < 1 || k.v > 2 || k.tq > 3) KT_FAIL("unsupported sampling factors");
                if (k.h > hmax) hmax = k.h;
                if (k.v > vmax) vmax = k.v;
                k.decoded = false;
                k.td = k.ta = 0;
                k.pred = 0;
            }
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (int c = 0; c < nc; ++c) {
                Component& k = comps[c];
                k.blocksW = mcux * k.h;
                k.blocksH = mcuy * k.v;
                k.width = (W * k.h + hmax - 1) / hmax;
                k.height = (H * k.v + vmax - 1) / vmax;
                k.plane.assign((size_t)k.blocksW * 8 * k.blocksH * 8, 128);
            }
            have_sof = true;
            break;
        }
        case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            KT_FAIL("progressive / lossless / arithmetic JPEG is not supported");
        case 0xDD:  // DRI
            if (n < 2) KT_FAIL("truncated DRI");
            restart_interval = (s[0] << 8) | s[1];
            break;
        case 0xDA: {  // SOS
            if (!have_sof) KT_FAIL("scan before frame header");
            if (n < 1) KT_FAIL("truncated SOS");
            const int ns = s[0];
            if (ns < 1 || ns > (int)comps.size() || n < (size_t)(1 + 2 * ns + 3)) KT_FAIL("bad SOS");
            int order[3];
            for (int i = 0; i < ns; ++i) {
                int ci = -1;
                for (size_t c = 0; c < comps.size(); ++c)
                    if (comps[c].id == s[1 + 2 * i]) ci = (int)c;
                if (ci < 0 || comps[ci].decoded) KT_FAIL("bad component in SOS");
                comps[ci].td = s[2 + 2 * i] >> 4;
                comps[ci].ta = s[2 + 2 * i] & 15;
                if (comps[ci].td > 3 || comps[ci].ta > 3 || !hdc[comps[ci].td].present || !hac[comps[ci].ta].present) KT_FAIL("missing Huffman table");
                if (!qt_present[comps[ci].tq]) KT_FAIL("missing quantisation table");
                comps[ci].pred = 0;
                order[i] = ci;
            }
            BitReader br(data, size);
            br.seek(p + len);
            // MCU grid of this scan: interleaved -> whole-image MCUs; a single component -> its own blocks (T.81 A.2.2 / A.2.3)
            const bool inter = ns > 1;
            int mx, my;
            if (inter) { mx = comps[0].blocksW / comps[0].h; my = comps[0].blocksH / comps[0].v; }
            else { mx = (comps[order[0]].width + 7) / 8; my = (comps[order[0]].height + 7) / 8; }
            int until_restart = restart_interval;
            int coef[64];
            for (int y = 0; y < my; ++y)
                for (int x = 0; x < mx; ++x) {
                    if (restart_interval && until_restart == 0) {
                        if (!br.restart()) KT_FAIL("missing restart marker");
                        for (int i = 0; i < ns; ++i) comps[order[i]].pred = 0;
                        until_restart = restart_interval;
                    }
                    for (int i = 0; i < ns; ++i) {
                        Component& k = comps[order[i]];
                        const int bh = inter ? k.h : 1, bv = inter ? k.v : 1;
                        for (int by = 0; by < bv; ++by)
                            for (int bx = 0; bx < bh; ++bx) {
                                std::memset(coef, 0, sizeof(coef));
                                const int t = decode_symbol(br, hdc[k.td]);
                                if (t < 0 || t > 11) KT_FAIL("corrupt DC code");
                                k.pred += extend(br.bits(t), t);
                                if (k.pred < -32768 || k.pred > 32767) KT_FAIL("corrupt DC coefficient");
                                coef[0] = k.pred;
                                for (int kk = 1; kk < 64;) {
                                    const int rs = decode_symbol(br, hac[k.ta]);
                                    if (rs < 0) KT_FAIL("corrupt AC code");
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r == 15) { kk += 16; continue; }
                                        break;  // EOB
                                    }
                                    kk += r;
                                    if (kk > 63 || sz > 10) KT_FAIL("corrupt AC coefficient");
                                    coef[kZigzag[kk]] = extend(br.bits(sz), sz);
                                    ++kk;
                                }
                                const int bxx = x * bh + bx, byy = y * bv + by;
                                if (bxx < k.blocksW && byy < k.blocksH)
                                    idct_islow(coef, qt[k.tq], &k.plane[((size_t)byy * 8) * ((size_t)k.blocksW * 8) + (size_t)bxx * 8], k.blocksW * 8, rl);
                            }
                    }
                    if (restart_interval) --until_restart;
                }
            for (int i = 0; i < ns; ++i) comps[order[i]].decoded = true;
            p = br.pos_bytes();
            bool all = true;
            for (size_t c = 0; c < comps.size(); ++c) all = all && comps[c].decoded;
            if (all) done = true;
            continue;  // p already points past the entropy-coded data
        }
        default: break;  // APPn, COM, ...: skipped
        }
        p += len;
    }
    if (!have_sof) KT_FAIL("no frame header");
    for (size_t c = 0; c < comps.size(); ++c)
        if (!comps[c].decoded) KT_FAIL("a component has no scan");

    // upsample every component to full size (jdsample.c), then colour-convert (jdcolor.c) into B G R
    std::vector<unsigned char> full[3];
    const unsigned char* rows[3];
    int strides[3];
    for (size_t c = 0; c < comps.size(); ++c) {
        Component& k = comps[c];
        const int stride = k.blocksW * 8;
        const int hr = hmax / k.h, vr = vmax / k.v;
        if (hr == 1 && vr == 1) { rows[c] = k.plane.data(); strides[c] = stride; continue; }
        const int ow = 2 * k.width + 2;
        full[c].assign((size_t)ow * (size_t)(H + 2), 0);
        strides[c] = ow;
        rows[c] = full[c].data();
        const bool fancy = k.width > 2;   // jinit_upsampler: the fancy routines need downsampled_width > 2
        for (int y = 0; y < H; ++y) {
            unsigned char* o = &full[c][(size_t)y * ow];
            const int sy = vr == 2 ? y >> 1 : y;
            const unsigned char* near = &k.plane[(size_t)(sy < k.height ? sy : k.height - 1) * stride];
            if (hr == 2 && vr == 1 && fancy) fancy_h2_row(near, k.width, o);
            else if (hr == 2 && vr == 2 && fancy) {
                // the upper output row of a pair leans on the input row above, the lower one on the row below; beyond the image the
                // nearest real row is repeated (jdmainct.c context rows)
                int fy = (y & 1) ? sy + 1 : sy - 1;
                if (fy < 0) fy = 0;
                if (fy > k.height - 1) fy = k.height - 1;
                fancy_h2v2_row(near, &k.plane[(size_t)fy * stride], k.width, o);
            } else {
                for (int x = 0; x < W; ++x) o[x] = near[hr == 2 ? x >> 1 : x];   // h2v1_upsample / h2v2_upsample / h1v2: replication
            }
        }
    }
    if (comps.size() == 1) {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const unsigned char g = rows[0][(size_t)y * strides[0] + x];
                unsigned char* o = bgr + ((size_t)y * W + x) * 3;
                o[0] = o[1] = o[2] = g;
            }
        return true;
    }
    // build_ycc_rgb_table / ycc_rgb_convert
    struct Ycc {   // FIX(1.40200), FIX(1.77200), FIX(0.71414), FIX(0.34414) at SCALEBITS = 16
        int crr[256], cbb[256], crg[256], cbg[256];
        Ycc()
        {
            for (int i = 0; i < 256; ++i) {
                const int x = i - 128;
                crr[i] = (91881 * x + 32768) >> 16;
                cbb[i] = (116130 * x + 32768) >> 16;
                crg[i] = -46802 * x;
                cbg[i] = -22554 * x + 32768;
            }
        }
    };
    static const Ycc ycc;
    const int *crr = ycc.crr, *cbb = ycc.cbb, *crg = ycc.crg, *cbg = ycc.cbg;
    for (int y = 0; y < H; ++y) {
        const unsigned char* py = rows[0] + (size_t)y * strides[0];
        const unsigned char* pb = rows[1] + (size_t)y * strides[1];
        const unsigned char* pr = rows[2] + (size_t)y * strides[2];
        unsigned char* o = bgr + (size_t)y * W * 3;
        for (int x = 0; x < W; ++x) {
            const int Y = py[x], cb = pb[x], cr = pr[x];
            int r = Y + crr[cr], g = Y + ((cbg[cb] + crg[cr]) >> 16), b = Y + cbb[cb];
            o[3 * x + 0] = (unsigned char)(b < 0 ? 0 : (b > 255 ? 255 : b));
            o[3 * x + 1] = (unsigned char)(g < 0 ? 0 : (g > 255 ? 255 : g));
            o[3 * x + 2] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
    return true;
#undef KT_FAIL
}

}  // namespace jpeg
}  // namespace kt
