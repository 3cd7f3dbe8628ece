// jpeg_tool -- decodes one JPEG file with the .klg colour decoder (JpegDecoder.h) and writes the raw BGR bytes, or encodes raw BGR bytes
// with the place-recognition sample encoder (JpegEncoder.h); used by tests/test_jpeg.py (no GPU needed).
//   jpeg_tool <in.jpg> <width> <height> <out.bgr>          jpeg_tool -e <in.bgr> <width> <height> <quality> <out.jpg>
//   jpeg_tool -pr <in.bgr> <width> <height> <out.jpg> <out.bgr>: the image (+ a synthetic depth map) through a PlaceRecognitionInput:
//   compress(), the JPEG it holds -> out.jpg, decompressImgTo -> out.bgr, decompressDepthTo checked against the input (exit code 3)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "JpegDecoder.h"
#include "JpegEncoder.h"
#include "PlaceRecognitionInput.h"

int main(int argc, char** argv)
{
    if (argc == 7 && std::string(argv[1]) == "-e") {
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), q = std::atoi(argv[5]);
        if (w <= 0 || h <= 0 || w > 16384 || h > 16384) { std::fprintf(stderr, "bad size\n"); return 2; }
        std::vector<unsigned char> img((size_t)w * h * 3), jpg;
        FILE* fi = std::fopen(argv[2], "rb");
        if (!fi || std::fread(img.data(), 1, img.size(), fi) != img.size()) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
        std::fclose(fi);
        if (!kt::jpeg::encodeBGR(img.data(), w, h, q, jpg)) { std::fprintf(stderr, "encode failed\n"); return 1; }
        FILE* fo = std::fopen(argv[6], "wb");
        if (!fo) return 2;
        std::fwrite(jpg.data(), 1, jpg.size(), fo);
        std::fclose(fo);
        return 0;
    }
    if (argc == 7 && std::string(argv[1]) == "-pr") {
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
        if (w <= 0 || h <= 0 || w > 16384 || h > 16384) { std::fprintf(stderr, "bad size\n"); return 2; }
        Resolution::get(w, h);
        const size_t n = (size_t)w * h;
        unsigned char* img = new unsigned char[n * 3];
        unsigned short* depth = new unsigned short[n];
        FILE* fi = std::fopen(argv[2], "rb");
        if (!fi || std::fread(img, 1, n * 3, fi) != n * 3) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
        std::fclose(fi);
        std::vector<unsigned short> depth0(n);
        for (size_t i = 0; i < n; ++i) depth0[i] = depth[i] = (unsigned short)(500 + (i * 7919u) % 3000u);
        PlaceRecognitionInput pr(img, (int)(n * 3), depth, (int)(n * 2), false, 1234, 5, kt::Vector3f(), kt::Matrix3f());
        pr.compress();
        if (!pr.isCompressed || pr.imageIsRaw || pr.imageSize <= 0 || pr.depthSize <= 0) return 3;
        FILE* fo = std::fopen(argv[5], "wb");
        if (!fo) return 2;
        std::fwrite(pr.rgbImage, 1, (size_t)pr.imageSize, fo);
        std::fclose(fo);
        std::vector<unsigned char> back(n * 3);
        std::vector<unsigned short> dback(n);
        pr.decompressImgTo(back.data());
        pr.decompressDepthTo((unsigned char*)dback.data());
        if (std::memcmp(dback.data(), depth0.data(), n * 2) != 0) return 3;
        fo = std::fopen(argv[6], "wb");
        if (!fo) return 2;
        std::fwrite(back.data(), 1, back.size(), fo);
        std::fclose(fo);
        return 0;
    }
    if (argc != 5) { std::fprintf(stderr, "usage: %s in.jpg width height out.bgr\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    std::vector<unsigned char> in;
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) in.insert(in.end(), buf, buf + n);
    std::fclose(f);
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384) { std::fprintf(stderr, "bad size\n"); return 2; }
    std::vector<unsigned char> out((size_t)w * h * 3);
    std::string err;
    if (!kt::jpeg::decodeBGR(in.data(), in.size(), w, h, out.data(), &err)) { std::fprintf(stderr, "decode failed: %s\n", err.c_str()); return 1; }
    f = std::fopen(argv[4], "wb");
    if (!f) return 2;
    std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return 0;
}
