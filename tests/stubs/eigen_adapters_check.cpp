// Compiles host/EigenAdapters.h against the stand-in headers of this directory and runs every adapter once (no GPU, no libkt_hip.so:
// the adapters are header-only memory-layout identities).
#include <cstdio>
#include <cstring>
#include "EigenAdapters.h"
#if !defined(KT_HAVE_EIGEN) || !defined(KT_HAVE_PCL)
#error "the stand-in headers were not picked up"
#endif
int main()
{
    kt::Matrix3f m;
    for (int k = 0; k < 9; ++k) m.data()[k] = (float)(k + 1);
    kt::EigenMatrix3fRM e = kt::toEigen(m);
    if (e(0, 1) != 2.0f || e(1, 0) != 4.0f || e(2, 2) != 9.0f) return 1;             // row-major on both sides
    kt::Matrix3f back = kt::fromEigen(e);
    if (std::memcmp(back.data(), m.data(), 36) != 0) return 2;
    kt::Vector3f v(1.5f, -2.5f, 3.5f);
    Eigen::Vector3f ev = kt::toEigen(v);
    kt::Vector3f vb = kt::fromEigen(ev);
    if (ev(1) != -2.5f || vb(2) != 3.5f) return 3;
    kt::Matrix4f p;
    p(0, 3) = 7.0f; p(3, 0) = -1.0f;
    Eigen::Matrix4f ep = kt::toEigen(p);
    if (ep(0, 3) != 7.0f || ep(3, 0) != -1.0f || ep.data()[12] != 7.0f) return 4;      // the Eigen side is column-major: (0, 3) is element 12
    Mat33& dm = kt::device_cast_mat33(e);
    float3& dv = kt::device_cast_float3(ev);
    if (dm.data[0].y != 2.0f || dm.data[2].z != 9.0f || dv.y != -2.5f) return 5;      // device_cast: the same 36 / 12 bytes
    PointXYZRGB pts[2];
    std::memset(pts, 0, sizeof(pts));
    pts[1].x = 4.0f; pts[1].r = 200; pts[1].a = 9;
    pcl::PointCloud<pcl::PointXYZRGB> cloud;
    kt::toPcl(pts, 2, cloud);
    if (cloud.width != 2 || cloud.height != 1 || cloud.points[1].x != 4.0f || cloud.points[1].r != 200 || cloud.points[1].a != 9) return 6;
    PointXYZRGBNormal np[1];
    std::memset(np, 0, sizeof(np));
    np[0].normal_z = -1.0f; np[0].curvature = 0.25f; np[0].g = 17;
    pcl::PointCloud<pcl::PointXYZRGBNormal> ncloud;
    kt::toPcl(np, 1, ncloud);
    if (ncloud.points[0].normal_z != -1.0f || ncloud.points[0].curvature != 0.25f || ncloud.points[0].g != 17) return 7;
    std::printf("eigen adapters ok\n");
    return 0;
}
