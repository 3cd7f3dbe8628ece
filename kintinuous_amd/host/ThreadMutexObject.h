// ThreadMutexObject.h -- a value guarded by its own mutex, with a condition variable to announce changes: the helper the tracker's
// shared fields are made of (interface of utils/ThreadMutexObject.h:26-138: assignValue, getValue, waitForSignal, ...).
// Built on the C++11 primitives (no boost in this image); every accessor goes through one private `guarded()` helper.
#pragma once

#include <stdint.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

template <class T>
class ThreadMutexObject {
    typedef std::lock_guard<std::mutex> Guard;

    T value_;                               // the shared value
    T snapshot_;                            // what the last getter returned (getReferenceWait hands out a reference to it)
    std::mutex lock_;
    std::condition_variable_any changed_;

    // run f(value_) under the lock
    template <class F> void guarded(F f) { Guard g(lock_); f(value_); }
    T& refresh() { Guard g(lock_); snapshot_ = value_; return snapshot_; }
    static void nap(int microseconds) { std::this_thread::sleep_for(std::chrono::microseconds(microseconds)); }

  public:
    ThreadMutexObject() : value_(), snapshot_() {}
    ThreadMutexObject(T initialValue) : value_(initialValue), snapshot_(initialValue) {}

    // writers
    void assignValue(T newValue) { Guard g(lock_); value_ = newValue; snapshot_ = newValue; }
    void assignAndNotifyAll(T newValue) { guarded([&](T& v) { v = newValue; changed_.notify_all(); }); }
    void notifyAll() { guarded([&](T&) { changed_.notify_all(); }); }
    void operator++(int) { guarded([](T& v) { v++; }); }
    void operator+=(const uint64_t& other) { guarded([&](T& v) { v += other; }); }

    // readers: a copy of the value as it is now / after the next notification / after `wait` microseconds
    T getValue() { return refresh(); }
    T waitForSignal()
    {
        std::unique_lock<std::mutex> ul(lock_);
        changed_.wait(ul);
        snapshot_ = value_;
        return snapshot_;
    }
    T getValueWait(int wait = 33000) { nap(wait); return refresh(); }
    T& getReferenceWait(int wait = 33000) { nap(wait); return refresh(); }

    // raw access for callers that hold getMutex() themselves
    std::mutex& getMutex() { return lock_; }
    T& getReference() { return value_; }
};
