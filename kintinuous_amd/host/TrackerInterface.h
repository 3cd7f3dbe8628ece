// TrackerInterface.h -- the reference's tracker thread (backend/TrackerInterface.h:32-82, TrackerInterface.cpp:20-137): a ThreadObject whose
// process() grabs a frame, uploads it and calls processFrame; at the end of the log (or when endRequested is raised, MainController::
// complete) it raises pauseCapture and finalised, finalises the frontend and keeps waking the CloudSliceProcessor until that thread
// reports cloudSliceProcessorFinished; with threadPack.limit set a frame takes at least 33 333 us.  One instance per GPU.
// The constructor takes the intrinsics as an Intr (the reference: a cv::Mat *; host/EigenAdapters.h converts) and one extra switch,
// operatorPath: compose every frame from the internal.h operators (the reference's own structure) instead of the device-resident tracker.
#pragma once

#include <unistd.h>
#include <algorithm>
#include <cstdio>

#include "KintinuousTracker.h"
#include "RawLogReader.h"
#include "ThreadObject.h"

class TrackerInterface : public ThreadObject {
  public:
    TrackerInterface(LogReader* logRead, const Intr& depthIntrinsics, bool operatorPath = false)
        : ThreadObject("TrackerInterfaceThread"), endRequested(false), logRead(logRead), currentFrame(0), firstRun(true), operatorPath(operatorPath),
          primed(false), haveNext(false), nextDepth(0), nextImage(0), nextTime(0)
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
    void loadTrajectory(const std::string& filename) { frontend->loadTrajectory(filename); }
    int getCurrentFrame() const { return currentFrame; }   // (shell addition: frames handed to the frontend so far)

    ThreadMutexObject<bool> endRequested;
    ThreadMutexObject<bool> handshakeTimedOut;   // (shell addition) the end-of-run hand-shake below gave up
    int handshakeSeconds = 0;                    // (shell addition) 0 = wait for ever, as the reference does (TrackerInterface.cpp:66-69); > 0: give up after that many seconds

  private:
    // one turn of ThreadObject::run()'s loop (TrackerInterface.cpp:44-137)
    bool inline process()
    {
        if (firstRun) {
            kt::device::context(ConfigArgs::get().gpu);   // cudaSetDevice on the tracker thread's first turn (:48)
            firstRun = false;
        }
        if (threadPack.pauseCapture.getValue()) {
            usleep(1000);   // (the reference spins on the flag; a paused tracker need not hold a core)
            return true;
        }
        TICK(threadIdentifier);
        const uint64_t start = Stopwatch::getCurrentSystemTime();
        bool returnVal = true;
        const bool shouldEnd = endRequested.getValue();
        if (shouldEnd || !grabFrame(returnVal)) {
            threadPack.pauseCapture.assignValue(true);
            threadPack.finalised.assignValue(true);
            finalise();
            // the FINAL slice is out: keep waking the slice processor until it has taken it over (:66-69)
            // (the reference loops for ever, napping 33 ms before every look; here the flag is looked at first -- a run without a processor
            // thread, or one that has finished already, pays no nap.  The loop is the reference's: it ends when the processor has taken the
            // FINAL slice, however long its backlog is (advisor, round 5: a 20 s default abandoned a processor that was merely behind); a
            // caller that prefers an error to a hang sets handshakeSeconds and checks handshakeTimedOut)
            const uint64_t waitStart = Stopwatch::getCurrentSystemTime();
            while (!threadPack.cloudSliceProcessorFinished.getValue()) {
                {
                    std::lock_guard<std::mutex> lock(frontend->cloudMutex);
                    frontend->cloudSignal.notify_all();
                }
                if (handshakeSeconds > 0 && Stopwatch::getCurrentSystemTime() - waitStart > (uint64_t)handshakeSeconds * 1000000ull) {
                    std::fprintf(stderr, "TrackerInterface: the slice processor did not take the FINAL slice over within %d s\n", handshakeSeconds);
                    handshakeTimedOut.assignValue(true);
                    break;
                }
                usleep(2000);
            }
            return shouldEnd ? false : returnVal;
        }
        trackFrame();
        const uint64_t duration = Stopwatch::getCurrentSystemTime() - start;
        if (threadPack.limit.getValue() && duration < 33333) {
            const int sleepTime = std::max(int(33333 - duration), 0);
            usleep(sleepTime);
        }
        TOCK(threadIdentifier);
        return true;
    }

    // logRead->grabNext for the frame this turn tracks.  On the device-resident path the log is read one frame ahead: frame k + 1 is
    // announced (pinned copy, upload, pose-independent stages on the tracker's second stream) before frame k is tracked.
    bool grabFrame(bool& returnVal)
    {
        if (operatorPath) return logRead->grabNext(returnVal, currentFrame);
        if (!primed) {
            primed = true;
            haveNext = logRead->grabNext(returnVal, currentFrame);
            if (haveNext) latchNext();
        }
        if (!haveNext) { returnVal = false; return false; }
        cur = next;
        haveNext = logRead->grabNext(returnVal, currentFrame);   // the reader rotates its frame buffers: cur's buffers stay valid
        if (haveNext) {
            latchNext();
            frontend->announceFrame(nextDepth, nextImage);
        }
        returnVal = true;
        return true;
    }

    void trackFrame()
    {
        ++currentFrame;
        threadPack.trackerFrame.assignAndNotifyAll(currentFrame);   // (RawLogReader.cpp:140 does this inside grabNext)
        if (operatorPath) {
            const int rows = Resolution::get().rows(), cols = Resolution::get().cols();
            depth_device.upload(logRead->decompressedDepth, (size_t)cols * 2, rows, cols);     // TrackerInterface.cpp:90-91
            colors_device.upload(logRead->decompressedImage, (size_t)cols * 3, rows, cols);
            TICK("processFrame");
            frontend->processFrame(depth_device, colors_device, logRead->decompressedImage, logRead->decompressedDepth,
                                   (uint64_t)logRead->timestamp, logRead->isCompressed, logRead->compressedDepth, logRead->compressedDepthSize,
                                   logRead->compressedImage, logRead->compressedImageSize);   // :93-102
            TOCK("processFrame");
            return;
        }
        frontend->processFrameHost(cur.depth, cur.image, cur.time, cur.compressed, cur.compDepth, cur.compDepthSize, cur.compImage, cur.compImageSize);
    }

    struct Latched {
        unsigned short* depth = 0;
        unsigned char* image = 0;
        uint64_t time = 0;
        bool compressed = false;
        unsigned char *compDepth = 0, *compImage = 0;
        int compDepthSize = 0, compImageSize = 0;
    };
    void latchNext()
    {
        nextDepth = logRead->decompressedDepth; nextImage = logRead->decompressedImage; nextTime = (uint64_t)logRead->timestamp;
        next.depth = nextDepth; next.image = nextImage; next.time = nextTime;
        next.compressed = logRead->isCompressed;
        next.compDepth = logRead->compressedDepth; next.compDepthSize = logRead->compressedDepthSize;
        next.compImage = logRead->compressedImage; next.compImageSize = logRead->compressedImageSize;
    }
    Latched cur, next;
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
