// tests/stubs/pcl/point_cloud.h -- NOT PCL: the members of pcl::PointCloud<T> host/EigenAdapters.h writes (points, width, height, is_dense).
#pragma once
#include <stdint.h>
#include <vector>
namespace pcl {
template <typename T>
struct PointCloud { std::vector<T> points; uint32_t width = 0, height = 0; bool is_dense = true; };
}  // namespace pcl
