// STAND-IN for the reference's own header (utils/ThreadObject.h): in an integration the reference's file is used as it is and this
// one is deleted.  It exists so that the shell and its tests build without boost; it is interface glue, not product code -- do not grow it.
//
// ThreadObject.h -- the base of every thread of the reference (utils/ThreadObject.h:26-97), restated on std:: primitives: start() runs
// process() until it returns false or stop() raises haltSignal; running() tells a controller whether the loop is still alive;
// threadPack is the shared ThreadDataPack; lagTime is what the GUI shows as "lag".  A controller starts one as
//     std::thread(&ThreadObject::start, component)          (MainController.cpp:146: boost::bind(&ThreadObject::start, ...))
#pragma once

#include <assert.h>
#include <stdio.h>
#include <string>

#include "Stopwatch.h"
#include "ThreadDataPack.h"

class ThreadObject {
  public:
    ThreadObject(std::string threadIdentifier) : threadPack(ThreadDataPack::get()), threadIdentifier(threadIdentifier)
    {
        // Heartbeat
        Stopwatch::get().pulse(threadIdentifier);
        Stopwatch::get().sendAll();
        haltSignal.assignValue(false);
        isRunning.assignValue(false);
        lagTime.assignValue(0);
    }

    virtual ~ThreadObject() {}

    virtual void reset() {}

    void stop() { haltSignal.assignValue(true); }

    void start()
    {
        haltSignal.assignValue(false);
        run();
    }

    std::string getThreadIdentifier() { return threadIdentifier; }

    bool running() { return isRunning.getValue(); }

    ThreadDataPack& threadPack;
    ThreadMutexObject<uint64_t> lagTime;

  protected:
    void run()
    {
        announce(" started\n");
        isRunning.assignValue(true);
        while (process() && !haltSignal.getValue()) Stopwatch::get().sendAll();
        isRunning.assignValue(false);
        announce(" ended\n");
    }

    // The reference's words (utils/ThreadObject.h:73-85), but each announcement leaves as ONE write: several ThreadObjects start at the
    // same moment and an unsynchronised std::cout interleaved them ("...Thread startedTracker...").
    void announce(const char* what)
    {
        std::string line = threadIdentifier + what;
        flockfile(stdout);                    // one locked stdio operation: ordered with every other printf / std::cout of the process
        fputs(line.c_str(), stdout);
        fflush(stdout);
        funlockfile(stdout);
    }

    virtual bool process()
    {
        assert(false);
        return false;
    }

    std::string threadIdentifier;
    ThreadMutexObject<bool> haltSignal;
    ThreadMutexObject<bool> isRunning;
};
