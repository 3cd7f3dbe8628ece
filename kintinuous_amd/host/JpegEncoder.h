// JpegEncoder.h -- BGR24 -> baseline JPEG, for PlaceRecognitionInput::compress() (frontend/PlaceRecognitionInput.h:208-224).
//
// The reference hands the colour image of a place-recognition sample to cvEncodeImage(".jpg", img, {CV_IMWRITE_JPEG_QUALITY, 90}), i.e.
// OpenCV 2.4's JpegEncoder -> the system libjpeg (libjpeg-turbo on Ubuntu 14.04 / 15.04, README.md:14-31) with the library defaults:
// BGR -> RGB swap, in_color_space = JCS_RGB, jpeg_set_defaults + jpeg_set_quality(90, TRUE): YCbCr, 4:2:0 (luma 2x2, chroma 1x1),
// the Annex K quantisation and Huffman tables (no optimisation pass), dct_method = JDCT_ISLOW, a JFIF 1.01 APP0, one interleaved scan.
// Neither library is in this image, so this is a restatement of those published algorithms (IJG libjpeg 6b / libjpeg-turbo: jcparam.c
// quality scaling, jccolor.c fixed-point RGB -> YCbCr, jcprepct.c / jcsample.c edge replication and the h2v2 box filter with its
// alternating bias, jfdctint.c "slow-but-accurate integer FDCT", jcdctmgr.c quantisation (round half away from zero), jccoefct.c dummy
// blocks, jchuff.c, jcmarker.c), written to give the same BYTES.  Pinned: tests/test_jpeg.py compares the stream byte for byte with
// libjpeg-turbo's own encoder (through Pillow, quality 90 and others, 4:2:0, ragged and MCU-aligned sizes).
#pragma once

#include <stdint.h>
#include <cstring>
#include <vector>

namespace kt {
namespace jpeg {

namespace enc_detail {

static const unsigned char kZigzagE[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                           41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                           30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// ITU T.81 Annex K.1 (natural order), jcparam.c std_luminance_quant_tbl / std_chrominance_quant_tbl
static const unsigned char kQLuma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                         69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55, 64,
                                         81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92, 95,  98,  112, 100, 103, 99};
static const unsigned char kQChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                           99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// Annex K.3 Huffman tables (jcparam.c std_huff_tables)
static const unsigned char kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const unsigned char kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const unsigned char kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const unsigned char kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const unsigned char kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
static const unsigned char kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const unsigned char kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

struct HuffCode {
    unsigned short code[256];
    unsigned char len[256];
    HuffCode(const unsigned char* bits, const unsigned char* vals, int nvals)   // jchuff.c jpeg_make_c_derived_tbl
    {
        std::memset(code, 0, sizeof(code));
        std::memset(len, 0, sizeof(len));
        unsigned int c = 0;
        int k = 0;
        for (int l = 1; l <= 16; ++l) {
            for (int i = 0; i < bits[l - 1] && k < nvals; ++i, ++k) { code[vals[k]] = (unsigned short)c++; len[vals[k]] = (unsigned char)l; }
            c <<= 1;
        }
    }
};

// jfdctint.c: jpeg_fdct_islow on an 8x8 block of centred samples; the result is scaled up by 8
inline void fdctIslow(int* data)
{
    const int CB = 13, P1 = 2;
    const int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
              F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
#define KT_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
    for (int pass = 0; pass < 2; ++pass) {
        for (int q = 0; q < 8; ++q) {
            int* d = pass == 0 ? data + 8 * q : data + q;
            const int st = pass == 0 ? 1 : 8;
            const int tmp0 = d[0] + d[7 * st], tmp7 = d[0] - d[7 * st], tmp1 = d[st] + d[6 * st], tmp6 = d[st] - d[6 * st];
            const int tmp2 = d[2 * st] + d[5 * st], tmp5 = d[2 * st] - d[5 * st], tmp3 = d[3 * st] + d[4 * st], tmp4 = d[3 * st] - d[4 * st];
            const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            if (pass == 0) {
                d[0] = (tmp10 + tmp11) << P1;
                d[4 * st] = (tmp10 - tmp11) << P1;
            } else {
                d[0] = KT_DESCALE(tmp10 + tmp11, P1);
                d[4 * st] = KT_DESCALE(tmp10 - tmp11, P1);
            }
            const int sh = pass == 0 ? CB - P1 : CB + P1;
            int z1 = (tmp12 + tmp13) * F0541;
            d[2 * st] = KT_DESCALE(z1 + tmp13 * F0765, sh);
            d[6 * st] = KT_DESCALE(z1 + tmp12 * (-F1847), sh);
            z1 = tmp4 + tmp7;
            int z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
            const int z5 = (z3 + z4) * F1175;
            const int t4 = tmp4 * F0298, t5 = tmp5 * F2053, t6 = tmp6 * F3072, t7 = tmp7 * F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            d[7 * st] = KT_DESCALE(t4 + z1 + z3, sh);
            d[5 * st] = KT_DESCALE(t5 + z2 + z4, sh);
            d[3 * st] = KT_DESCALE(t6 + z2 + z3, sh);
            d[st] = KT_DESCALE(t7 + z1 + z4, sh);
        }
    }
#undef KT_DESCALE
}

struct BitWriter {   // jchuff.c emit_bits / flush_bits: MSB first, 0xFF stuffed with 0x00, the last byte padded with ones
    std::vector<unsigned char>& out;
    unsigned int acc;
    int n;
    explicit BitWriter(std::vector<unsigned char>& o) : out(o), acc(0), n(0) {}
    void put(unsigned int code, int len)
    {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const unsigned char b = (unsigned char)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);
            n -= 8;
        }
    }
    void flush() { put(0x7F, 7); acc = 0; n = 0; }
};

inline int bitLength(int v)
{
    int nb = 0;
    while (v) { ++nb; v >>= 1; }
    return nb;
}

// jchuff.c encode_one_block
inline void encodeBlock(BitWriter& bw, const short* blk, int& lastDc, const HuffCode& dc, const HuffCode& ac)
{
    int temp = blk[0] - lastDc, temp2 = temp;
    lastDc = blk[0];
    if (temp < 0) { temp = -temp; --temp2; }
    int nbits = bitLength(temp);
    bw.put(dc.code[nbits], dc.len[nbits]);
    if (nbits) bw.put((unsigned int)temp2, nbits);
    int r = 0;
    for (int k = 1; k < 64; ++k) {
        temp = blk[kZigzagE[k]];
        if (temp == 0) { ++r; continue; }
        while (r > 15) { bw.put(ac.code[0xF0], ac.len[0xF0]); r -= 16; }
        temp2 = temp;
        if (temp < 0) { temp = -temp; --temp2; }
        nbits = bitLength(temp);
        const int sym = (r << 4) + nbits;
        bw.put(ac.code[sym], ac.len[sym]);
        bw.put((unsigned int)temp2, nbits);
        r = 0;
    }
    if (r > 0) bw.put(ac.code[0], ac.len[0]);
}

inline void marker(std::vector<unsigned char>& o, int m, int len)
{
    o.push_back(0xFF); o.push_back((unsigned char)m);
    o.push_back((unsigned char)(len >> 8)); o.push_back((unsigned char)(len & 0xFF));
}

}  // namespace enc_detail

// cvEncodeImage(".jpg", bgr, {CV_IMWRITE_JPEG_QUALITY, quality}): the bytes libjpeg writes for this image with its defaults (see above).
inline bool encodeBGR(const unsigned char* bgr, int cols, int rows, int quality, std::vector<unsigned char>& out)
{
    using namespace enc_detail;
    out.clear();
    if (!bgr || cols <= 0 || rows <= 0 || cols > 65535 || rows > 65535) return false;
    // jcparam.c jpeg_quality_scaling + jpeg_add_quant_table (force_baseline)
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    unsigned char qt[2][64];
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) {
            long v = ((long)(t == 0 ? kQLuma[i] : kQChroma[i]) * scale + 50L) / 100L;
            if (v <= 0) v = 1;
            if (v > 255) v = 255;
            qt[t][i] = (unsigned char)v;
        }
    // geometry (jcmaster.c initial_setup / per_scan_setup): luma 2x2, chroma 1x1, MCU = 16 x 16 pixels
    const int mcusX = (cols + 15) / 16, mcusY = (rows + 15) / 16;
    const int wbY = (cols + 7) / 8, hbY = (rows + 7) / 8;                       // width / height in blocks: ceil(size * samp / (max * 8))
    const int wbC = (cols + 15) / 16, hbC = (rows + 15) / 16;
    const int rows2 = (rows + 1) & ~1;                                         // jcprepct.c: the conversion buffer is padded to max_v_samp rows
    const int planeW = mcusX * 16, planeH = mcusY * 16;                         // luma plane incl. every padding; chroma: half of both
    std::vector<unsigned char> Y((size_t)planeW * planeH), Cb((size_t)planeW * planeH), Cr((size_t)planeW * planeH);
    // jccolor.c rgb_ycc_convert (SCALEBITS = 16)
    const int FIX_299 = 19595, FIX_587 = 38470, FIX_114 = 7471, FIX_16874 = 11059, FIX_33126 = 21709, FIX_5 = 32768, FIX_41869 = 27439,
              FIX_08131 = 5329, ONE_HALF = 1 << 15, CBCR = 128 << 16;
    for (int y = 0; y < rows; ++y) {
        const unsigned char* p = bgr + (size_t)y * cols * 3;
        unsigned char* py = &Y[(size_t)y * planeW];
        unsigned char* pb = &Cb[(size_t)y * planeW];
        unsigned char* pr = &Cr[(size_t)y * planeW];
        for (int x = 0; x < cols; ++x) {
            const int b = p[3 * x], g = p[3 * x + 1], r = p[3 * x + 2];
            py[x] = (unsigned char)((FIX_299 * r + FIX_587 * g + FIX_114 * b + ONE_HALF) >> 16);
            pb[x] = (unsigned char)((-FIX_16874 * r - FIX_33126 * g + FIX_5 * b + CBCR + ONE_HALF - 1) >> 16);
            pr[x] = (unsigned char)((FIX_5 * r - FIX_41869 * g - FIX_08131 * b + CBCR + ONE_HALF - 1) >> 16);
        }
    }
    // jcprepct.c expand_bottom_edge on the colour-converted rows (to an even number of rows)
    for (int y = rows; y < rows2; ++y)
        for (std::vector<unsigned char>* pl : {&Y, &Cb, &Cr}) std::memcpy(&(*pl)[(size_t)y * planeW], &(*pl)[(size_t)(rows - 1) * planeW], (size_t)cols);
    // jcsample.c: luma = fullsize_downsample (copy + expand_right_edge to width_in_blocks * 8); chroma = h2v2_downsample after
    // expand_right_edge to 2 * width_in_blocks * 8, bias 1, 2, 1, 2, ...
    std::vector<unsigned char> cb((size_t)(planeW / 2) * (planeH / 2)), cr((size_t)(planeW / 2) * (planeH / 2));
    for (int y = 0; y < rows2; ++y) {
        for (std::vector<unsigned char>* pl : {&Y, &Cb, &Cr}) {
            unsigned char* row = &(*pl)[(size_t)y * planeW];
            const int need = pl == &Y ? wbY * 8 : wbC * 16;
            for (int x = cols; x < need; ++x) row[x] = row[cols - 1];
        }
    }
    for (int y = 0; y < rows2 / 2; ++y)
        for (int pass = 0; pass < 2; ++pass) {
            const unsigned char* r0 = &(pass == 0 ? Cb : Cr)[(size_t)(2 * y) * planeW];
            const unsigned char* r1 = r0 + planeW;
            unsigned char* o = &(pass == 0 ? cb : cr)[(size_t)y * (planeW / 2)];
            int bias = 1;
            for (int x = 0; x < wbC * 8; ++x) {
                o[x] = (unsigned char)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + bias) >> 2);
                bias ^= 3;
            }
        }
    // jcprepct.c: the downsampled rows are padded to a whole iMCU row by repeating the last one
    for (int y = rows2; y < planeH; ++y) std::memcpy(&Y[(size_t)y * planeW], &Y[(size_t)(rows2 - 1) * planeW], (size_t)wbY * 8);
    for (int y = rows2 / 2; y < planeH / 2; ++y) {
        std::memcpy(&cb[(size_t)y * (planeW / 2)], &cb[(size_t)(rows2 / 2 - 1) * (planeW / 2)], (size_t)wbC * 8);
        std::memcpy(&cr[(size_t)y * (planeW / 2)], &cr[(size_t)(rows2 / 2 - 1) * (planeW / 2)], (size_t)wbC * 8);
    }

    // ---- markers (jcmarker.c: write_file_header, write_frame_header, write_scan_header)
    out.reserve((size_t)cols * rows / 4 + 1024);
    out.push_back(0xFF); out.push_back(0xD8);
    marker(out, 0xE0, 16);
    const unsigned char jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    out.insert(out.end(), jfif, jfif + 14);
    for (int t = 0; t < 2; ++t) {
        marker(out, 0xDB, 67);
        out.push_back((unsigned char)t);
        for (int i = 0; i < 64; ++i) out.push_back(qt[t][kZigzagE[i]]);
    }
    marker(out, 0xC0, 17);
    out.push_back(8);
    out.push_back((unsigned char)(rows >> 8)); out.push_back((unsigned char)(rows & 0xFF));
    out.push_back((unsigned char)(cols >> 8)); out.push_back((unsigned char)(cols & 0xFF));
    out.push_back(3);
    const unsigned char comps[9] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};
    out.insert(out.end(), comps, comps + 9);
    struct Dht { int id; const unsigned char* bits; const unsigned char* vals; int n; };
    const Dht dht[4] = {{0x00, kDcLumaBits, kDcVals, 12}, {0x10, kAcLumaBits, kAcLumaVals, 162}, {0x01, kDcChromaBits, kDcVals, 12},
                        {0x11, kAcChromaBits, kAcChromaVals, 162}};
    for (const Dht& h : dht) {
        marker(out, 0xC4, 2 + 1 + 16 + h.n);
        out.push_back((unsigned char)h.id);
        out.insert(out.end(), h.bits, h.bits + 16);
        out.insert(out.end(), h.vals, h.vals + h.n);
    }
    marker(out, 0xDA, 12);
    const unsigned char sos[10] = {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
    out.insert(out.end(), sos, sos + 10);

    // ---- entropy-coded data (jccoefct.c compress_data + jcdctmgr.c forward_DCT + jchuff.c)
    const HuffCode dcL(kDcLumaBits, kDcVals, 12), acL(kAcLumaBits, kAcLumaVals, 162), dcC(kDcChromaBits, kDcVals, 12), acC(kAcChromaBits, kAcChromaVals, 162);
    BitWriter bw(out);
    int lastDc[3] = {0, 0, 0};
    short mcu[6][64];
    auto forward = [&](const unsigned char* plane, int stride, int bx, int by, const unsigned char* q, short* coef) {
        int ws[64];
        for (int y = 0; y < 8; ++y) {
            const unsigned char* s = plane + (size_t)(by * 8 + y) * stride + bx * 8;
            for (int x = 0; x < 8; ++x) ws[y * 8 + x] = (int)s[x] - 128;
        }
        fdctIslow(ws);
        for (int i = 0; i < 64; ++i) {
            const int qval = (int)q[i] << 3;
            int t = ws[i];
            if (t < 0) { t = -t; t += qval >> 1; t = t >= qval ? t / qval : 0; t = -t; }
            else { t += qval >> 1; t = t >= qval ? t / qval : 0; }
            coef[i] = (short)t;
        }
    };
    for (int my = 0; my < mcusY; ++my)
        for (int mx = 0; mx < mcusX; ++mx) {
            int blkn = 0;
            // luma: 2 x 2 blocks; blocks past the component's width / height in blocks are dummies (zero AC, DC of the block before)
            for (int yi = 0; yi < 2; ++yi)
                for (int xi = 0; xi < 2; ++xi, ++blkn) {
                    const int bx = mx * 2 + xi, by = my * 2 + yi;
                    if (by < hbY && bx < wbY) forward(Y.data(), planeW, bx, by, qt[0], mcu[blkn]);
                    else { std::memset(mcu[blkn], 0, sizeof(mcu[blkn])); mcu[blkn][0] = mcu[blkn - 1][0]; }
                }
            for (int c = 0; c < 2; ++c, ++blkn) {
                if (my < hbC && mx < wbC) forward(c == 0 ? cb.data() : cr.data(), planeW / 2, mx, my, qt[1], mcu[blkn]);
                else { std::memset(mcu[blkn], 0, sizeof(mcu[blkn])); mcu[blkn][0] = mcu[blkn - 1][0]; }
            }
            for (int b = 0; b < 4; ++b) encodeBlock(bw, mcu[b], lastDc[0], dcL, acL);
            encodeBlock(bw, mcu[4], lastDc[1], dcC, acC);
            encodeBlock(bw, mcu[5], lastDc[2], dcC, acC);
        }
    bw.flush();
    out.push_back(0xFF); out.push_back(0xD9);
    return true;
}

}  // namespace jpeg
}  // namespace kt
