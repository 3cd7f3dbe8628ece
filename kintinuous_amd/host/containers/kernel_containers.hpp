// kernel_containers.hpp -- the four pointer views the reference's operator signatures are written in (interface of
// frontend/cuda/containers/kernel_containers.hpp:49-92): DevPtr (pointer), PtrSz (+ element count), PtrStep (+ row stride in bytes),
// PtrStepSz (+ rows and columns).  They own nothing; DeviceArray / DeviceArray2D convert to them implicitly (device_array.hpp).
#pragma once

#include <cstddef>
#include <type_traits>

template <typename T>
struct DevPtr {
    typedef T elem_type;
    const static size_t elem_size = sizeof(T);

    T* data;

    DevPtr(T* p = 0) : data(p) {}
    size_t elemSize() const { return sizeof(T); }
    operator T*() { return data; }
    operator const T*() const { return data; }
};

template <typename T>
struct PtrSz : DevPtr<T> {
    size_t size;   // elements

    PtrSz() : DevPtr<T>(0), size(0) {}
    PtrSz(T* p, size_t count) : DevPtr<T>(p), size(count) {}
};

template <typename T>
struct PtrStep : DevPtr<T> {
    size_t step;   // bytes from one row to the next

    PtrStep() : DevPtr<T>(0), step(0) {}
    PtrStep(T* p, size_t row_bytes) : DevPtr<T>(p), step(row_bytes) {}

    // first element of row y
    T* ptr(int y = 0) { return row(this->data, y); }
    const T* ptr(int y = 0) const { return row(this->data, y); }

  private:
    template <typename P> P* row(P* base, int y) const
    {
        typedef typename std::conditional<std::is_const<P>::value, const char, char>::type Byte;
        return reinterpret_cast<P*>(reinterpret_cast<Byte*>(base) + (size_t)y * step);
    }
};

template <typename T>
struct PtrStepSz : PtrStep<T> {
    int cols, rows;

    PtrStepSz() : PtrStep<T>(), cols(0), rows(0) {}
    PtrStepSz(int nrows, int ncols, T* p, size_t row_bytes) : PtrStep<T>(p, row_bytes), cols(ncols), rows(nrows) {}
};
