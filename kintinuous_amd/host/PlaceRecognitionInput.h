// PlaceRecognitionInput.h -- one frame sampled for the loop-closure backend (frontend/PlaceRecognitionInput.h:27-214): the image and
// depth bytes as the log delivered them (raw, or zlib depth + JPEG colour), the frame's time stamps and the camera pose of the sample.
// compress(): the depth goes through zlib's compress2 at Z_BEST_SPEED and the colour image through a baseline JPEG encoder at quality
// 90, like the reference's cvEncodeImage(".jpg", ..., {CV_IMWRITE_JPEG_QUALITY, 90}) (JpegEncoder.h restates libjpeg's default
// compression path: the bytes are libjpeg-turbo's, tests/test_jpeg.py) -- sequentially here, on two boost threads there.  A sample that
// still holds raw pixels says so in `imageIsRaw`.  decompressImgTo / decompressDepthTo undo what the log (or compress()) did.
#pragma once

#include <stdint.h>
#include <zlib.h>
#include <cassert>
#include <cstring>

#include "ConfigArgs.h"
#include "JpegDecoder.h"
#include "JpegEncoder.h"
#include "LinearAlgebra.h"
#include "Resolution.h"

class PlaceRecognitionInput {
  public:
    PlaceRecognitionInput(unsigned char* rgbImage, int imageSize, unsigned short* depthMap, int depthSize, bool isCompressed, uint64_t utime,
                          uint64_t lagTime, const kt::Vector3f& trans, const kt::Matrix3f& rotation)
        : rgbImage(rgbImage), imageSize(imageSize), depthMap(depthMap), depthSize(depthSize), isCompressed(isCompressed),
          originallyCompressed(isCompressed), imageIsRaw(!isCompressed), utime(utime), lagTime(lagTime), trans(trans), rotation(rotation)
    {
    }
    PlaceRecognitionInput()
        : rgbImage(0), imageSize(0), depthMap(0), depthSize(0), isCompressed(false), originallyCompressed(false), imageIsRaw(true), utime(0), lagTime(0)
    {
    }
    virtual ~PlaceRecognitionInput() { dump(); }

    void compress()  // :72-118
    {
        assert(!isCompressed);
        unsigned long compressed_size = (unsigned long)Resolution::get().numPixels() * sizeof(int16_t) * 4;
        uint8_t* buf = new uint8_t[compressed_size];
        compress2(buf, &compressed_size, (const Bytef*)depthMap, (unsigned long)depthSize, Z_BEST_SPEED);
        uint8_t* tmp = new uint8_t[compressed_size];
        std::memcpy(tmp, buf, compressed_size);
        delete[] buf;
        delete[] depthMap;
        depthMap = (unsigned short*)tmp;
        depthSize = (int)compressed_size;
        if (imageIsRaw && rgbImage) {  // encodeJpeg, :208-224
            std::vector<unsigned char> jpg;
            if (kt::jpeg::encodeBGR(rgbImage, Resolution::get().cols(), Resolution::get().rows(), 90, jpg)) {
                unsigned char* img = new unsigned char[jpg.size()];
                std::memcpy(img, jpg.data(), jpg.size());
                delete[] rgbImage;
                rgbImage = img;
                imageSize = (int)jpg.size();
                imageIsRaw = false;
            }
        }
        isCompressed = true;
    }
    void decompressImgTo(unsigned char* target)  // :120-133
    {
        assert(isCompressed);
        const int cols = Resolution::get().cols(), rows = Resolution::get().rows();
        if (imageIsRaw) { std::memcpy(target, rgbImage, (size_t)cols * rows * 3); return; }
        std::string err;
        if (!kt::jpeg::decodeBGR(rgbImage, (size_t)imageSize, cols, rows, target, &err)) std::memset(target, 0, (size_t)cols * rows * 3);
        else if (ConfigArgs::get().flipColors && originallyCompressed)
            for (int i = 0; i < cols * rows; ++i) { const unsigned char t = target[3 * i]; target[3 * i] = target[3 * i + 2]; target[3 * i + 2] = t; }
    }
    void decompressDepthTo(unsigned char* target)  // :135-140
    {
        assert(isCompressed);
        unsigned long n = (unsigned long)Resolution::get().numPixels() * 2;
        uncompress(target, &n, (const Bytef*)depthMap, (unsigned long)depthSize);
    }
    void dump()  // :142-187
    {
        delete[] rgbImage;
        rgbImage = 0;
        imageSize = 0;
        delete[] depthMap;
        depthMap = 0;
        depthSize = 0;
        isCompressed = false;
        imageIsRaw = true;
        utime = 0;
        lagTime = 0;
        trans = kt::Vector3f();
        rotation = kt::Matrix3f();
    }

    unsigned char* rgbImage;
    int imageSize;
    unsigned short* depthMap;
    int depthSize;
    bool isCompressed;
    bool originallyCompressed;
    bool imageIsRaw;
    uint64_t utime;
    uint64_t lagTime;
    kt::Vector3f trans;
    kt::Matrix3f rotation;

  private:
    PlaceRecognitionInput(const PlaceRecognitionInput&);
    PlaceRecognitionInput& operator=(const PlaceRecognitionInput&);
};
