// ThreadObject.h -- the base of every thread of the reference (utils/ThreadObject.h:26-97), restated on std:: primitives: start() runs
// process() until it returns false or stop() raises haltSignal; running() tells a controller whether the loop is still alive;
// threadPack is the shared ThreadDataPack; lagTime is what the GUI shows as "lag".  A controller starts one as
//     std::thread(&ThreadObject::start, component)          (MainController.cpp:146: boost::bind(&ThreadObject::start, ...))
#pragma once

#include <assert.h>
#include <iostream>
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
        std::cout << threadIdentifier << " started" << std::endl;
        isRunning.assignValue(true);
        while (process() && !haltSignal.getValue()) Stopwatch::get().sendAll();
        isRunning.assignValue(false);
        std::cout << threadIdentifier << " ended" << std::endl;
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
