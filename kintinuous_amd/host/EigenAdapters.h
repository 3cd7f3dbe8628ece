// EigenAdapters.h -- conversions between the shell's plain linear-algebra structs (LinearAlgebra.h) and the Eigen / PCL types the
// reference's backend passes around.  Compiled only where those libraries exist; this build image has
// neither -- tests/test_host_logic.py::test_eigen_adapters_type_check compiles and runs every function against stand-in headers with
// the shapes of those types (tests/stubs/).  Every function is a memory-layout identity (the structs were laid out to match:
// row-major 3x3 floats, 3 floats, 32-byte PointXYZRGB, 48-byte PointXYZRGBNormal).
#pragma once

#include "LinearAlgebra.h"
#include "internal.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#define KT_HAVE_EIGEN 1
namespace kt {
typedef Eigen::Matrix<float, 3, 3, Eigen::RowMajor> EigenMatrix3fRM;   // the reference's rotation type (KintinuousTracker.h:206)
inline EigenMatrix3fRM toEigen(const Matrix3f& m) { return Eigen::Map<const EigenMatrix3fRM>(m.data()); }
inline Eigen::Vector3f toEigen(const Vector3f& v) { return Eigen::Map<const Eigen::Vector3f>(v.data()); }
inline Matrix3f fromEigen(const EigenMatrix3fRM& m) { Matrix3f r; Eigen::Map<EigenMatrix3fRM>(r.data()) = m; return r; }
inline Vector3f fromEigen(const Eigen::Vector3f& v) { return Vector3f(v(0), v(1), v(2)); }
// DensePose::pose (KintinuousTracker.h:60-70) is a column-major Eigen::Matrix4f in the reference; the shell keeps it row-major
inline Eigen::Matrix4f toEigen(const Matrix4f& m)
{
    Eigen::Matrix4f r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r(i, j) = m(i, j);
    return r;
}
// device_cast<Mat33>(Eigen row-major 3x3), internal.h:478-482: the same 36 bytes
inline Mat33& device_cast_mat33(EigenMatrix3fRM& m) { return *reinterpret_cast<Mat33*>(m.data()); }
inline float3& device_cast_float3(Eigen::Vector3f& v) { return *reinterpret_cast<float3*>(v.data()); }
}  // namespace kt
#endif

#if __has_include(<pcl/point_types.h>)
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#define KT_HAVE_PCL 1
namespace kt {
static_assert(sizeof(pcl::PointXYZRGB) == sizeof(PointXYZRGB), "pcl::PointXYZRGB layout");
static_assert(sizeof(pcl::PointXYZRGBNormal) == sizeof(PointXYZRGBNormal), "pcl::PointXYZRGBNormal layout");
// CloudSlice::cloud / processedCloud (CloudSlice.h:28-129) as PCL clouds: a byte copy, the records have PCL's layout
inline void toPcl(const PointXYZRGB* pts, size_t n, pcl::PointCloud<pcl::PointXYZRGB>& out)
{
    out.points.resize(n);
    if (n) std::memcpy(&out.points[0], pts, n * sizeof(PointXYZRGB));
    out.width = (uint32_t)n; out.height = 1; out.is_dense = true;
}
inline void toPcl(const PointXYZRGBNormal* pts, size_t n, pcl::PointCloud<pcl::PointXYZRGBNormal>& out)
{
    out.points.resize(n);
    if (n) std::memcpy(&out.points[0], pts, n * sizeof(PointXYZRGBNormal));
    out.width = (uint32_t)n; out.height = 1; out.is_dense = true;
}
}  // namespace kt
#endif
#endif
