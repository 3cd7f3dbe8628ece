// TrackerInterface.h -- the frame pump of the reference's tracker thread (backend/TrackerInterface.h:34-88,
// TrackerInterface.cpp:20-137) without the threading shell: grab a frame, upload, processFrame; finalise at the end of
// the log.  One instance per GPU.
#pragma once

#include "KintinuousTracker.h"
#include "RawLogReader.h"

class TrackerInterface {
  public:
    TrackerInterface(LogReader* logRead, const Intr& depthIntrinsics, bool operatorPath = false)
        : logRead(logRead), currentFrame(0), firstRun(true), operatorPath(operatorPath), primed(false), haveNext(false), nextDepth(0),
          nextImage(0), nextTime(0)
    {
        kt::device::context(ConfigArgs::get().gpu);  // cudaSetDevice(ConfigArgs::get().gpu), TrackerInterface.cpp:48
        frontend = new KintinuousTracker(depthIntrinsics, operatorPath);
        reset();
    }
    virtual ~TrackerInterface() { delete frontend; }

    void reset() { currentFrame = 0; primed = false; haveNext = false; frontend->reset(); }
    KintinuousTracker* getFrontend() { return frontend; }
    void finalise() { frontend->finalise(); }
    void setPark(const bool park) { frontend->setParked(park); }
    void enableOverlap() { frontend->setOverlap(2); }
    int getCurrentFrame() const { return currentFrame; }

    // one iteration of TrackerInterface::process(): false once the log is exhausted (after finalise()).  On the device-resident
    // path the log is read one frame ahead: frame k + 1 is announced (pinned copy, upload, pose-independent stages on the tracker's
    // second stream) before frame k is tracked.
    bool process()
    {
        bool returnVal = true;
        if (operatorPath) {
            if (!logRead->grabNext(returnVal, currentFrame)) { finalise(); return false; }
            ++currentFrame;
            const int rows = Resolution::get().rows(), cols = Resolution::get().cols();
            depth_device.upload(logRead->decompressedDepth, (size_t)cols * 2, rows, cols);
            colors_device.upload(logRead->decompressedImage, (size_t)cols * 3, rows, cols);
            frontend->processFrame(depth_device, colors_device, logRead->decompressedImage, logRead->decompressedDepth,
                                   (uint64_t)logRead->timestamp, logRead->isCompressed, logRead->compressedDepth, logRead->compressedDepthSize,
                                   logRead->compressedImage, logRead->compressedImageSize);
            return true;
        }
        if (!primed) {
            primed = true;
            haveNext = logRead->grabNext(returnVal, currentFrame);
            if (haveNext) latchNext();
        }
        if (!haveNext) { finalise(); return false; }
        unsigned short* depth = nextDepth;
        unsigned char* image = nextImage;
        const uint64_t time = nextTime;
        const bool comp = nextCompressed;
        unsigned char *cd = nextCompDepth, *ci = nextCompImage;
        const int cds = nextCompDepthSize, cis = nextCompImageSize;
        haveNext = logRead->grabNext(returnVal, currentFrame);   // the reader rotates its frame buffers: `depth` / `image` stay valid
        if (haveNext) {
            latchNext();
            frontend->announceFrame(nextDepth, nextImage);
        }
        ++currentFrame;
        frontend->processFrameHost(depth, image, time, comp, cd, cds, ci, cis);   // TrackerInterface.cpp:93-102
        return true;
    }

  private:
    void latchNext()
    {
        nextDepth = logRead->decompressedDepth; nextImage = logRead->decompressedImage; nextTime = (uint64_t)logRead->timestamp;
        nextCompressed = logRead->isCompressed;
        nextCompDepth = logRead->compressedDepth; nextCompDepthSize = logRead->compressedDepthSize;
        nextCompImage = logRead->compressedImage; nextCompImageSize = logRead->compressedImageSize;
    }
    bool nextCompressed = false;
    unsigned char *nextCompDepth = 0, *nextCompImage = 0;
    int nextCompDepthSize = 0, nextCompImageSize = 0;
    LogReader* logRead;
    KintinuousTracker* frontend;
    DeviceArray2D<unsigned short> depth_device;
    DeviceArray2D<PixelRGB> colors_device;
    int currentFrame;
    bool firstRun, operatorPath, primed, haveNext;
    unsigned short* nextDepth;
    unsigned char* nextImage;
    uint64_t nextTime;
};
