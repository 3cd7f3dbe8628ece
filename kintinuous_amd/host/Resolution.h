// Resolution.h -- the image size every part of the shell agrees on (interface of frontend/Resolution.h:24-68).
// Set once: the first get(width, height) fixes it for the process, later calls take no arguments.
#pragma once

#include <cstdio>
#include <cstdlib>

class Resolution {
    struct Size { int w, h, n; };
    Size sz;
    explicit Resolution(Size s) : sz(s) {}

    static Size checked(int w, int h)
    {
        if (w <= 0 || h <= 0) {
            std::fprintf(stderr, "Resolution: get(width, height) must be called with the image size before anything asks for it\n");
            std::abort();
        }
        Size s = {w, h, w * h};
        return s;
    }

  public:
    static const Resolution& get(int width = 0, int height = 0)
    {
        static const Resolution the_one(checked(width, height));
        return the_one;
    }
    // width == cols, height == rows; numPixels == cols * rows
    const int& cols() const { return sz.w; }
    const int& rows() const { return sz.h; }
    const int& width() const { return sz.w; }
    const int& height() const { return sz.h; }
    const int& numPixels() const { return sz.n; }
};
