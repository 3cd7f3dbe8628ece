// RawLogReader.h -- .klg reader (utils/LogReader.h, utils/RawLogReader.cpp:21-150): int32 numFrames, then per frame
// int64 timestamp, int32 depthSize, int32 imageSize, depth bytes, image bytes.  Depth: raw u16 (depthSize == 2*W*H) or a
// zlib stream; image: raw rgb24 (imageSize == 3*W*H), absent (imageSize == 0 -> zeros) or a JPEG stream (cvDecodeImage in the
// reference, JpegDecoder.h here).  hasMore() keeps the reference's off-by-one: the last frame of a log is never returned.
//
// Decode-ahead (`-dt <threads>`, ConfigArgs::decodeThreads; default: a quarter of the hardware threads, at most 8; 0 = decode inside
// grabNext like the reference): a compressed VGA frame costs ~3 ms of JPEG decoding and ~2 ms of inflate on one core, the GPU path
// consumes a frame in 0.35 ms.  With N worker threads the
// records are still read from the file strictly in order (one reader at a time), decoded in parallel into a ring of frame slots and
// handed out in order; what grabNext returns -- buffers, sizes, isCompressed, the points at which a corrupt or truncated log stops
// the run -- is the same in both modes (tests/test_jpeg.py compares them frame by frame).
#pragma once

#include <stdint.h>
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
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
    explicit RawLogReader(const std::string& file = ConfigArgs::get().logFile, int decodeThreads = ConfigArgs::get().decodeThreads)
        : fp(0), numFrames(0), currentFrame(0), cols(Resolution::get().width()), rows(Resolution::get().height()),
          flipColors(ConfigArgs::get().flipColors), nextToRead(0), stopping(false), inputEnded(false)
    {
        fp = std::fopen(file.c_str(), "rb");
        if (!fp) { std::fprintf(stderr, "cannot open log %s\n", file.c_str()); std::exit(1); }
        if (decodeThreads < 0) {   // not given: a quarter of the hardware threads, 1..8 (0 = the reference's synchronous reader)
            const unsigned hw = std::thread::hardware_concurrency();
            decodeThreads = hw / 4 < 1 ? 1 : (hw / 4 > 8 ? 8 : (int)(hw / 4));
        }
        if (decodeThreads > 64) decodeThreads = 64;
        // a frame handed out stays valid for kKeep - 1 further grabNext calls (the tracker's read-ahead keeps up to two earlier frames
        // alive by address); the workers may run 2 frames per thread ahead of the consumer
        ring.resize((size_t)(kKeep + 2 * decodeThreads));
        const int n = Resolution::get().numPixels();
        for (size_t k = 0; k < ring.size(); ++k) {
            ring[k].depth.resize(n);
            ring[k].image.resize((size_t)n * 3);
        }
        decompressedDepth = ring[0].depth.data();
        decompressedImage = ring[0].image.data();
        int32_t frames = 0;
        if (std::fread(&frames, sizeof(int32_t), 1, fp) != 1) frames = 0;
        numFrames = frames;
        for (int k = 0; k < decodeThreads; ++k) workers.push_back(std::thread(&RawLogReader::work, this));
    }
    virtual ~RawLogReader()
    {
        stopWorkers();
        if (fp) std::fclose(fp);
    }

    bool hasMore() const { return currentFrame + 1 < numFrames; }  // RawLogReader.cpp:147-150
    int getNumFrames() const { return numFrames; }

    bool grabNext(bool& returnVal, int& /*frame*/)
    {
        if (!hasMore()) { returnVal = false; return false; }
        Slot* s = &ring[(size_t)currentFrame % ring.size()];
        if (workers.empty()) {
            if (!readRecord(*s, currentFrame)) s->state = Slot::ENDED;
            else s->state = decode(*s) ? Slot::READY : Slot::BAD;
        } else {
            std::unique_lock<std::mutex> lock(m);
            wake.notify_all();   // the consumer has moved on: one more slot may be filled
            done.wait(lock, [&] { return s->index == currentFrame && s->state >= Slot::READY; });
        }
        if (s->state == Slot::ENDED) { returnVal = false; return false; }   // a truncated file ends the log
        if (s->state == Slot::BAD) {   // a log is untrusted input: stop at the frame that is corrupt, having delivered the ones before it
            std::fprintf(stderr, "%s\n", s->error.c_str());
            stopWorkers();   // exit() runs the static destructors: no decoder may still be at work
            std::exit(1);
        }
        decompressedDepth = s->depth.data();
        decompressedImage = s->image.data();
        compressedDepth = s->rawDepth.data(); compressedDepthSize = s->depthSize;
        compressedImage = s->rawImage.data(); compressedImageSize = s->imageSize;
        timestamp = s->timestamp;
        isCompressed = s->compressed;
        {
            std::lock_guard<std::mutex> lock(m);
            ++currentFrame;
        }
        wake.notify_all();
        returnVal = true;
        return true;
    }

  private:
    void stopWorkers()
    {
        {
            std::lock_guard<std::mutex> lock(m);
            stopping = true;
        }
        wake.notify_all();
        for (size_t k = 0; k < workers.size(); ++k) workers[k].join();
        workers.clear();
    }

    struct Slot {
        enum State { EMPTY, BUSY, READY, ENDED, BAD };
        Slot() : timestamp(0), depthSize(0), imageSize(0), compressed(false), index(-1), state(EMPTY) {}
        std::vector<unsigned short> depth;
        std::vector<unsigned char> image, rawDepth, rawImage;
        int64_t timestamp;
        int32_t depthSize, imageSize;
        bool compressed;
        int index;     // the frame this slot holds or is being filled with
        State state;
        std::string error;
    };

    // the next record of the file into the slot's raw buffers (file order: one caller at a time).  false: the file ends here.
    bool readRecord(Slot& s, int frame)
    {
        const size_t n = (size_t)cols * (size_t)rows;
        s.index = frame;
        s.error.clear();
        int32_t depthSize = 0, imageSize = 0;
        if (std::fread(&s.timestamp, sizeof(int64_t), 1, fp) != 1 || std::fread(&depthSize, sizeof(int32_t), 1, fp) != 1 ||
            std::fread(&imageSize, sizeof(int32_t), 1, fp) != 1) return false;
        // payload sizes beyond any frame this resolution could produce mean a corrupt header
        const int32_t limit = (int32_t)(n * 16 + 65536);
        s.depthSize = depthSize; s.imageSize = imageSize;
        if (depthSize < 0 || imageSize < 0 || depthSize > limit || imageSize > limit) {
            char msg[160];
            std::snprintf(msg, sizeof(msg), "corrupt frame header in frame %d (payload sizes %d / %d)", frame, depthSize, imageSize);
            s.error = msg;
            s.depthSize = s.imageSize = 0;
            s.rawDepth.clear(); s.rawImage.clear();
            return true;   // decode() reports it
        }
        s.rawDepth.resize((size_t)depthSize);
        s.rawImage.resize((size_t)imageSize);
        if (depthSize > 0 && std::fread(s.rawDepth.data(), (size_t)depthSize, 1, fp) != 1) return false;
        if (imageSize > 0 && std::fread(s.rawImage.data(), (size_t)imageSize, 1, fp) != 1) return false;
        return true;
    }

    // raw payloads -> depth / image of the slot (touches nothing but the slot and the reader's constants).  false: s.error says why the frame is unusable.
    bool decode(Slot& s) const
    {
        if (!s.error.empty()) return false;
        const size_t n = (size_t)cols * (size_t)rows;
        const int32_t depthSize = s.depthSize, imageSize = s.imageSize;
        char msg[200];
        // the image decides isCompressed (RawLogReader.cpp:73-97); the depth payload has to agree (:99-117, asserts there)
        if ((size_t)imageSize == n * 3) {
            s.compressed = false;
            std::memcpy(s.image.data(), s.rawImage.data(), n * 3);
        } else if (imageSize > 0) {  // anything else is handed to cvDecodeImage -> B G R bytes
            s.compressed = true;
            std::string err;
            if (!kt::jpeg::decodeBGR(s.rawImage.data(), (size_t)imageSize, cols, rows, s.image.data(), &err)) {
                std::snprintf(msg, sizeof(msg), "cannot decode the colour image of frame %d: %s", s.index, err.c_str());
                s.error = msg;
                return false;
            }
        } else {
            s.compressed = false;
            std::memset(s.image.data(), 0, n * 3);
        }
        if ((size_t)depthSize == n * 2) {
            if (s.compressed) {
                std::snprintf(msg, sizeof(msg), "frame %d: raw depth with a compressed image", s.index);
                s.error = msg;
                return false;
            }
            std::memcpy(s.depth.data(), s.rawDepth.data(), n * 2);
        } else if (depthSize > 0) {
            // (the reference asserts isCompressed here; logs with zlib depth and a raw or empty image are accepted, depth decoded as is)
            uLongf decomp = (uLongf)(n * 2);
            if (uncompress((Bytef*)s.depth.data(), &decomp, (const Bytef*)s.rawDepth.data(), (uLong)depthSize) != Z_OK || decomp != (uLongf)(n * 2)) {
                // (a short stream would leave the tail of the slot holding an earlier frame)
                std::snprintf(msg, sizeof(msg), "corrupt zlib depth in frame %d (inflates to %lu of %zu bytes)", s.index, (unsigned long)decomp, n * 2);
                s.error = msg;
                return false;
            }
            if ((size_t)imageSize != n * 3 && imageSize > 0) s.compressed = true;
        } else {
            s.compressed = false;
            std::memset(s.depth.data(), 0, n * 2);
        }
        if (flipColors)  // RawLogReader.cpp:118-121 (cv::cvtColor RGB2BGR)
            for (size_t i = 0; i < n; ++i) { unsigned char t = s.image[i * 3]; s.image[i * 3] = s.image[i * 3 + 2]; s.image[i * 3 + 2] = t; }
        return true;
    }

    // worker: claim the next frame of the file while its slot is free, read it under the lock (file order), decode it outside
    void work()
    {
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            // Frame g lives in slot g % R and evicts frame g - R.  currentFrame = 1 + the frame handed out last; that frame and the
            // kKeep - 1 before it stay untouched -- exactly the lifetime of the synchronous reader (-dt 0), whose ring of kKeep slots
            // overwrites frame f when frame f + kKeep is read: a frame survives kKeep - 1 further grabNext calls in both modes.
            wake.wait(lock, [&] { return stopping || inputEnded || nextToRead + 1 >= numFrames || nextToRead < currentFrame - kKeep + (int)ring.size(); });
            if (stopping || inputEnded || nextToRead + 1 >= numFrames) return;
            const int g = nextToRead++;
            Slot& s = ring[(size_t)g % ring.size()];
            s.state = Slot::BUSY;
            const bool got = readRecord(s, g);
            if (got && !s.error.empty()) inputEnded = true;   // a corrupt header: the byte stream behind it cannot be framed either
            if (!got) {
                inputEnded = true;   // nothing behind a truncated record can be framed
                s.state = Slot::ENDED;
                done.notify_all();
                wake.notify_all();
                return;
            }
            lock.unlock();
            const bool ok = decode(s);
            lock.lock();
            s.state = ok ? Slot::READY : Slot::BAD;
            done.notify_all();
        }
    }

    FILE* fp;
    int numFrames, currentFrame;
    const int cols, rows;       // fixed at construction: the workers never ask the singletons
    const bool flipColors;
    static const int kKeep = 4;
    std::vector<Slot> ring;
    // decode-ahead state, all under m
    std::mutex m;
    std::condition_variable wake, done;   // wake: a slot came free / stop;  done: a slot became READY / ENDED / BAD
    std::vector<std::thread> workers;
    int nextToRead;
    bool stopping, inputEnded;
};
