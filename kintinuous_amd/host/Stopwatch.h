// Stopwatch.h -- the clock of utils/Stopwatch.h (getCurrentSystemTime: microseconds of gettimeofday, :122-127) and its TICK / TOCK
// markers.  The reference's Stopwatch broadcasts its timings over UDP to an external viewer (:90-120); that beacon is not part of this
// path, so the markers and pulse() / sendAll() are accepted and do nothing.
#pragma once

#include <stdint.h>
#include <sys/time.h>
#include <string>

#define TICK(name) ((void)0)
#define TOCK(name) ((void)0)

class Stopwatch {
  public:
    static Stopwatch& get()
    {
        static Stopwatch instance;
        return instance;
    }
    static Stopwatch& getInstance() { return get(); }
    static uint64_t getCurrentSystemTime()
    {
        timeval timeOfDay;
        gettimeofday(&timeOfDay, 0);
        return (uint64_t)timeOfDay.tv_sec * 1000000 + (uint64_t)timeOfDay.tv_usec;
    }
    void setCustomSignature(uint64_t) {}
    void pulse(const std::string&) {}
    void sendAll() {}
};
