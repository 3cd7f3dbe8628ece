// RawLogReader.h -- .klg reader (utils/LogReader.h, utils/RawLogReader.cpp:21-150): int32 numFrames, then per frame
// int64 timestamp, int32 depthSize, int32 imageSize, depth bytes, image bytes.  Depth: raw u16 (depthSize == 2*W*H) or a
// zlib stream; image: raw rgb24 (imageSize == 3*W*H), absent (imageSize == 0 -> zeros) or a JPEG stream (cvDecodeImage in the
// reference, JpegDecoder.h here).  hasMore() keeps the reference's off-by-one: the last frame of a log is never returned.
#pragma once

#include <stdint.h>
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ConfigArgs.h"
#include "JpegDecoder.h"
#include "Resolution.h"

class LogReader {
  public:
    LogReader() : decompressedDepth(0), decompressedImage(0), compressedDepth(0), compressedImage(0), compressedDepthSize(0), compressedImageSize(0),
                  timestamp(0), isCompressed(false) {}
    virtual ~LogReader() {}
    virtual bool grabNext(bool& returnVal, int& currentFrame) = 0;
    unsigned short* decompressedDepth;
    unsigned char* decompressedImage;
    unsigned char* compressedDepth;      // the frame's payloads as stored in the log (LogReader.h:51-54): what the place-recognition tap keeps
    unsigned char* compressedImage;
    int32_t compressedDepthSize;
    int32_t compressedImageSize;
    int64_t timestamp;
    bool isCompressed;
};

class RawLogReader : public LogReader {
  public:
    explicit RawLogReader(const std::string& file = ConfigArgs::get().logFile) : fp(0), numFrames(0), currentFrame(0)
    {
        fp = std::fopen(file.c_str(), "rb");
        if (!fp) { std::fprintf(stderr, "cannot open log %s\n", file.c_str()); std::exit(1); }
        const int n = Resolution::get().numPixels();
        for (int k = 0; k < kBuffers; ++k) {
            depthBuffers[k].resize(n);
            imageBuffers[k].resize((size_t)n * 3);
        }
        flip = 0;
        decompressedDepth = depthBuffers[0].data();
        decompressedImage = imageBuffers[0].data();
        int32_t frames = 0;
        if (std::fread(&frames, sizeof(int32_t), 1, fp) != 1) frames = 0;
        numFrames = frames;
    }
    virtual ~RawLogReader() { if (fp) std::fclose(fp); }

    bool hasMore() const { return currentFrame + 1 < numFrames; }  // RawLogReader.cpp:147-150
    int getNumFrames() const { return numFrames; }

    bool grabNext(bool& returnVal, int& /*frame*/)
    {
        if (!hasMore()) { returnVal = false; return false; }
        const size_t n = (size_t)Resolution::get().numPixels();
        // rotating frame buffers: the tracker's read-ahead keeps up to two earlier frames alive by address
        flip = (flip + 1) % kBuffers;
        std::vector<unsigned short>& depthBuffer = depthBuffers[flip];
        std::vector<unsigned char>& imageBuffer = imageBuffers[flip];
        decompressedDepth = depthBuffer.data();
        decompressedImage = imageBuffer.data();
        int32_t depthSize = 0, imageSize = 0;
        if (std::fread(&timestamp, sizeof(int64_t), 1, fp) != 1 || std::fread(&depthSize, sizeof(int32_t), 1, fp) != 1 ||
            std::fread(&imageSize, sizeof(int32_t), 1, fp) != 1) { returnVal = false; return false; }
        // a log is untrusted input: payload sizes beyond any frame this resolution could produce mean a corrupt header
        const int32_t limit = (int32_t)(n * 16 + 65536);
        if (depthSize < 0 || imageSize < 0 || depthSize > limit || imageSize > limit) {
            std::fprintf(stderr, "corrupt frame header in frame %d (payload sizes %d / %d)\n", currentFrame, depthSize, imageSize);
            std::exit(1);
        }
        std::vector<unsigned char>& rawDepth = rawDepthBuffers[flip];
        std::vector<unsigned char>& rawImage = rawImageBuffers[flip];
        rawDepth.resize((size_t)depthSize);
        rawImage.resize((size_t)imageSize);
        if (depthSize > 0 && std::fread(rawDepth.data(), (size_t)depthSize, 1, fp) != 1) { returnVal = false; return false; }
        if (imageSize > 0 && std::fread(rawImage.data(), (size_t)imageSize, 1, fp) != 1) { returnVal = false; return false; }
        compressedDepth = rawDepth.data(); compressedDepthSize = depthSize;
        compressedImage = rawImage.data(); compressedImageSize = imageSize;
        // the image decides isCompressed (RawLogReader.cpp:73-97); the depth payload has to agree (:99-117, asserts there)
        if ((size_t)imageSize == n * 3) {
            isCompressed = false;
            std::memcpy(imageBuffer.data(), rawImage.data(), n * 3);
        } else if (imageSize > 0) {  // anything else is handed to cvDecodeImage -> B G R bytes
            isCompressed = true;
            std::string err;
            if (!kt::jpeg::decodeBGR(rawImage.data(), (size_t)imageSize, Resolution::get().width(), Resolution::get().height(), imageBuffer.data(), &err)) {
                std::fprintf(stderr, "cannot decode the colour image of frame %d: %s\n", currentFrame, err.c_str());
                std::exit(1);
            }
        } else {
            isCompressed = false;
            std::memset(imageBuffer.data(), 0, n * 3);
        }
        if ((size_t)depthSize == n * 2) {
            if (isCompressed) { std::fprintf(stderr, "frame %d: raw depth with a compressed image\n", currentFrame); std::exit(1); }
            std::memcpy(depthBuffer.data(), rawDepth.data(), n * 2);
        } else if (depthSize > 0) {
            // (the reference asserts isCompressed here; logs with zlib depth and a raw or empty image are accepted, depth decoded as is)
            uLongf decomp = (uLongf)(n * 2);
            if (uncompress((Bytef*)depthBuffer.data(), &decomp, (const Bytef*)rawDepth.data(), (uLong)depthSize) != Z_OK || decomp != (uLongf)(n * 2)) {
                std::fprintf(stderr, "corrupt zlib depth in frame %d (inflates to %lu of %zu bytes)\n", currentFrame, (unsigned long)decomp, n * 2);
                std::exit(1);   // a short stream would leave the tail of the rotating buffer holding a frame from 4 reads ago
            }
            if ((size_t)imageSize != n * 3 && imageSize > 0) isCompressed = true;
        } else {
            isCompressed = false;
            std::memset(depthBuffer.data(), 0, n * 2);
        }
        if (ConfigArgs::get().flipColors)  // RawLogReader.cpp:118-121 (cv::cvtColor RGB2BGR)
            for (size_t i = 0; i < n; ++i) { unsigned char t = imageBuffer[i * 3]; imageBuffer[i * 3] = imageBuffer[i * 3 + 2]; imageBuffer[i * 3 + 2] = t; }
        ++currentFrame;
        returnVal = true;
        return true;
    }

  private:
    FILE* fp;
    int numFrames, currentFrame;
    static const int kBuffers = 4;
    int flip;
    std::vector<unsigned short> depthBuffers[kBuffers];
    std::vector<unsigned char> imageBuffers[kBuffers], rawDepthBuffers[kBuffers], rawImageBuffers[kBuffers];
};
