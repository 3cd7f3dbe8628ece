// ThreadMutexObject.h -- the value-behind-a-mutex helper the tracker's public fields are made of (utils/ThreadMutexObject.h:26-138),
// on std::mutex / std::condition_variable_any instead of the boost types (no boost in this image); same member functions.
#pragma once

#include <stdint.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

template <class T>
class ThreadMutexObject {
  public:
    ThreadMutexObject() : object(), lastCopy() {}
    ThreadMutexObject(T initialValue) : object(initialValue), lastCopy(initialValue) {}

    void assignValue(T newValue)
    {
        std::lock_guard<std::mutex> lock(mutex);
        object = lastCopy = newValue;
    }
    std::mutex& getMutex() { return mutex; }
    T& getReference() { return object; }
    void assignAndNotifyAll(T newValue)
    {
        std::lock_guard<std::mutex> lock(mutex);
        object = newValue;
        signal.notify_all();
    }
    void notifyAll()
    {
        std::lock_guard<std::mutex> lock(mutex);
        signal.notify_all();
    }
    T getValue()
    {
        std::lock_guard<std::mutex> lock(mutex);
        lastCopy = object;
        return lastCopy;
    }
    T waitForSignal()
    {
        std::unique_lock<std::mutex> lock(mutex);
        signal.wait(lock);
        lastCopy = object;
        return lastCopy;
    }
    T getValueWait(int wait = 33000)
    {
        std::this_thread::sleep_for(std::chrono::microseconds(wait));
        std::lock_guard<std::mutex> lock(mutex);
        lastCopy = object;
        return lastCopy;
    }
    T& getReferenceWait(int wait = 33000)
    {
        std::this_thread::sleep_for(std::chrono::microseconds(wait));
        std::lock_guard<std::mutex> lock(mutex);
        lastCopy = object;
        return lastCopy;
    }
    void operator++(int)
    {
        std::lock_guard<std::mutex> lock(mutex);
        object++;
    }
    void operator+=(const uint64_t& other)
    {
        std::lock_guard<std::mutex> lock(mutex);
        object += other;
    }

  private:
    T object;
    T lastCopy;
    std::mutex mutex;
    std::condition_variable_any signal;
};
