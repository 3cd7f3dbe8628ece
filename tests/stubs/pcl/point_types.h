// tests/stubs/pcl/point_types.h -- NOT PCL: the byte layouts of pcl::PointXYZRGB (32 bytes: float data[4]; then {b, g, r, a} in a
// 16-byte block) and pcl::PointXYZRGBNormal (48 bytes: data[4], data_n[4], {rgba, curvature, 8 bytes of padding}) as PCL 1.7's
// point_types.hpp declares them (PCL_ADD_POINT4D, PCL_ADD_NORMAL4D, PCL_ADD_RGB), for type-checking host/EigenAdapters.h.
#pragma once
#include <stdint.h>
namespace pcl {
struct alignas(16) PointXYZRGB { float x, y, z, pad0; union { struct { uint8_t b, g, r, a; }; float rgb; }; float pad1[3]; };
struct alignas(16) PointXYZRGBNormal { float x, y, z, pad0; float normal_x, normal_y, normal_z, pad1; union { struct { uint8_t b, g, r, a; }; float rgb; }; float curvature; float pad2[2]; };
}  // namespace pcl
