// The ThreadObject / ThreadMutexObject / ThreadDataPack protocol of the shell (host/ThreadObject.h = utils/ThreadObject.h:26-97) without a
// GPU: a worker that counts, started the way MainController starts its components (MainController.cpp:146) and stopped the way
// MainController::tearDown does (:188-233: stop(), then join).  tests/test_host_logic.py::test_thread_object_protocol compiles and runs it.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include "ThreadObject.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

class Counter : public ThreadObject {
  public:
    Counter(int limit) : ThreadObject("CounterThread"), count(0), limit(limit) {}
    std::atomic<int> count;

  private:
    bool process()
    {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        lagTime.assignValue((uint64_t)count.load());
        threadPack.latestPoseId.assignAndNotifyAll(++count);
        return limit < 0 || count.load() < limit;   // false ends the loop on its own
    }
    int limit;
};

static void sleep_ms(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

int main()
{
    ThreadDataPack& pack = ThreadDataPack::get();
    pack.reset();
    CHECK(&pack == &ThreadDataPack::get());                       // one pack per process
    CHECK(pack.limit.getValue() && !pack.finalised.getValue() && !pack.pauseCapture.getValue() && pack.latestPoseId.getValue() == 0);
    {
        Counter endless(-1);
        CHECK(&endless.threadPack == &pack && endless.getThreadIdentifier() == "CounterThread");
        CHECK(!endless.running());
        std::thread th(&ThreadObject::start, &endless);
        for (int k = 0; k < 2000 && !endless.running(); ++k) sleep_ms(1);
        CHECK(endless.running());
        const int seen = pack.latestPoseId.waitForSignal();          // a consumer waiting on the tracker's signal wakes up
        CHECK(seen >= 1);
        for (int k = 0; k < 2000 && endless.count.load() < 5; ++k) sleep_ms(1);
        CHECK(endless.count.load() >= 5 && endless.lagTime.getValue() >= 4);
        endless.stop();                                               // haltSignal: the loop ends after the step in flight
        th.join();
        CHECK(!endless.running());
        const int stopped_at = endless.count.load();
        sleep_ms(10);
        CHECK(endless.count.load() == stopped_at);
        std::thread again(&ThreadObject::start, &endless);            // start() lowers haltSignal again (reset + restart, MainController::reset)
        for (int k = 0; k < 2000 && endless.count.load() < stopped_at + 3; ++k) sleep_ms(1);
        CHECK(endless.count.load() >= stopped_at + 3);
        endless.stop();
        again.join();
        CHECK(!endless.running());
    }
    {
        Counter bounded(7);                                           // process() returning false ends the thread without stop()
        std::thread th(&ThreadObject::start, &bounded);
        th.join();
        CHECK(bounded.count.load() == 7 && !bounded.running());
    }
    // the end-of-run hand-shake the tracker thread and the slice processor play (TrackerInterface.cpp:96-112, CloudSliceProcessor.cpp:57-66)
    pack.reset();
    std::thread consumer([&pack]() {
        while (!pack.finalised.getValue()) pack.latestPoseId.waitForSignal();
        pack.cloudSliceProcessorFinished.assignValue(true);
    });
    sleep_ms(5);
    pack.finalised.assignValue(true);
    int spins = 0;
    while (!pack.cloudSliceProcessorFinished.getValue() && spins++ < 5000) { pack.notifyVariables(); sleep_ms(1); }
    consumer.join();
    CHECK(pack.cloudSliceProcessorFinished.getValue());
    std::printf("thread object ok\n");
    return 0;
}
