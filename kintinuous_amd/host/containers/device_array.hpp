// device_array.hpp -- DeviceArray<T> / DeviceArray2D<T> over the C-ABI (kt_malloc / kt_free / kt_upload / kt_download).
// Mirrors the public surface of the reference's containers (frontend/cuda/containers/device_array.hpp:60-255,
// device_memory.hpp:51-245): create / release / upload / download / ptr / cols / rows / empty, ref-counted shallow
// copies.  Differences by design: memory is dense (step == cols * sizeof(T); the reference's volume kernels already
// assume that, tsdf_volume.cu:612), and every allocation belongs to the process-wide kt_ctx of kt::device.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../../../include/kt_abi.h"
#include "kernel_containers.hpp"

namespace kt {

// cudaSafeCall / ___cudaSafeCall (internal.h:76-86): print and stop.  The reference exits on any device error.
inline void safeCall(int status, const char* file, int line)
{
    if (status == KT_OK) return;
    std::fprintf(stderr, "%s\t%s:%d\n", kt_last_error(), file, line);
    std::exit(1);
}
#define ktSafeCall(expr) ::kt::safeCall((expr), __FILE__, __LINE__)
// kt_tracker_num_*: a negative count reports a failed frame the same way
inline int count(int n)
{
    if (n >= 0) return n;
    std::fprintf(stderr, "%s\n", kt_last_error());
    std::exit(1);
}

// the implicit "current device" of the reference (cudaSetDevice in TrackerInterface.cpp:48) made explicit
class device {
  public:
    static kt_ctx* context(int gpu = -1)
    {
        static kt_ctx* ctx = nullptr;
        if (!ctx) ktSafeCall(kt_ctx_create(gpu < 0 ? 0 : gpu, &ctx));
        return ctx;
    }
    static void sync() { ktSafeCall(kt_sync(context())); }
};

}  // namespace kt

template <class T>
class DeviceArray {
  public:
    typedef T type;
    enum { elem_size = sizeof(T) };

    DeviceArray() : size_(0) {}
    explicit DeviceArray(size_t size) : size_(0) { create(size); }
    // non-owning view of user memory (DeviceArray(T* ptr, size_t size), device_array.hpp:82)
    DeviceArray(T* ptr, size_t size) : view_(ptr), size_(size) {}

    void create(size_t size)
    {
        if (mem_ && size == size_) return;
        void* p = nullptr;
        ktSafeCall(kt_malloc(kt::device::context(), size * sizeof(T), &p));
        mem_.reset(p, [](void* q) { kt_free(kt::device::context(), q); });
        view_ = nullptr;
        size_ = size;
    }
    void release() { mem_.reset(); view_ = nullptr; size_ = 0; }

    void upload(const T* host_ptr, size_t size)
    {
        create(size);
        ktSafeCall(kt_upload(kt::device::context(), ptr(), host_ptr, size * sizeof(T)));
    }
    void download(T* host_ptr) const { ktSafeCall(kt_download(kt::device::context(), host_ptr, ptr(), size_ * sizeof(T))); }
    void upload(const std::vector<T>& data) { upload(data.data(), data.size()); }
    void download(std::vector<T>& data) const { data.resize(size_); if (size_) download(data.data()); }

    T* ptr() { return view_ ? view_ : static_cast<T*>(mem_.get()); }
    const T* ptr() const { return view_ ? view_ : static_cast<const T*>(mem_.get()); }
    operator T*() { return ptr(); }
    operator const T*() const { return ptr(); }
    size_t size() const { return size_; }
    size_t sizeBytes() const { return size_ * sizeof(T); }
    bool empty() const { return ptr() == nullptr; }
    // DeviceMemory -> PtrSz<U> for any U (device_memory_impl.hpp:44-50)
    template <class U> operator PtrSz<U>() const { return PtrSz<U>((U*)ptr(), size_ * sizeof(T) / sizeof(U)); }

  private:
    std::shared_ptr<void> mem_;
    T* view_ = nullptr;
    size_t size_;
};

template <class T>
class DeviceArray2D {
  public:
    typedef T type;
    enum { elem_size = sizeof(T) };

    DeviceArray2D() : rows_(0), cols_(0) {}
    DeviceArray2D(int rows, int cols) : rows_(0), cols_(0) { create(rows, cols); }

    void create(int rows, int cols)
    {
        if (mem_ && rows == rows_ && cols == cols_) return;
        void* p = nullptr;
        const size_t bytes = (size_t)rows * cols * sizeof(T);
        ktSafeCall(kt_malloc(kt::device::context(), bytes, &p));
        // kinfu's maps rely on zero-initialised planes behind the NaN flag: make that defined
        ktSafeCall(kt_memset(kt::device::context(), p, 0, bytes));
        ktSafeCall(kt_sync(kt::device::context()));
        mem_.reset(p, [](void* q) { kt_free(kt::device::context(), q); });
        rows_ = rows;
        cols_ = cols;
    }
    void release() { mem_.reset(); rows_ = cols_ = 0; }

    // host_step in bytes, like DeviceMemory2D::upload (device_memory.cpp:206-215)
    void upload(const void* host_ptr, size_t host_step, int rows, int cols)
    {
        create(rows, cols);
        ktSafeCall(kt_upload2d(kt::device::context(), ptr(), host_ptr, host_step, (size_t)cols * sizeof(T), rows));
    }
    void download(void* host_ptr, size_t host_step) const
    {
        ktSafeCall(kt_download2d(kt::device::context(), host_ptr, host_step, ptr(), (size_t)cols_ * sizeof(T), rows_));
    }
    void upload(const std::vector<T>& data, int cols) { upload(data.data(), cols * sizeof(T), (int)(data.size() / cols), cols); }
    void download(std::vector<T>& data, int& cols) const
    {
        data.resize((size_t)rows_ * cols_);
        cols = cols_;
        if (!data.empty()) download(data.data(), cols_ * sizeof(T));
    }

    T* ptr(int y = 0) { return static_cast<T*>(mem_.get()) + (size_t)y * cols_; }
    const T* ptr(int y = 0) const { return static_cast<const T*>(mem_.get()) + (size_t)y * cols_; }
    operator T*() { return ptr(); }
    operator const T*() const { return ptr(); }
    int cols() const { return cols_; }
    int rows() const { return rows_; }
    size_t step() const { return (size_t)cols_ * sizeof(T); }
    bool empty() const { return !mem_; }
    // DeviceMemory2D -> PtrStep<U> / PtrStepSz<U> for ANY U (device_memory_impl.hpp:56-72): how the reference passes its
    // DeviceArray2D<int> colour volume as PtrStep<uchar4> and DeviceArray2D<PixelRGB> as PtrStepSz<uchar3>
    template <class U> operator PtrStep<U>() const { return PtrStep<U>((U*)ptr(), step()); }
    template <class U> operator PtrStepSz<U>() const { return PtrStepSz<U>(rows_, (int)(step() / sizeof(U)), (U*)ptr(), step()); }

  private:
    std::shared_ptr<void> mem_;
    int rows_, cols_;
};
