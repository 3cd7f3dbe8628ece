// jpeg_tool -- decodes one JPEG file with the .klg colour decoder (JpegDecoder.h) and writes the raw BGR bytes; used by
// tests/test_jpeg.py (no GPU needed).   jpeg_tool <in.jpg> <width> <height> <out.bgr>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "JpegDecoder.h"

int main(int argc, char** argv)
{
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
