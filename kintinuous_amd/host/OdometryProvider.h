// OdometryProvider.h -- frame-to-model odometry interface (frontend/OdometryProvider.h:33-69).
#pragma once

#include <stdint.h>
#include <cstring>

#include "CloudSlice.h"
#include "LinearAlgebra.h"
#include "internal.h"

// device_cast (internal.h:478-482) by value: host matrix / vector -> the operator API's POD types
namespace kt {
inline Mat33 dev(const Matrix3f& m) { Mat33 r; std::memcpy(&r, m.m, sizeof(r)); return r; }
inline float3 dev(const Vector3f& v) { float3 r = {v(0), v(1), v(2)}; return r; }
}  // namespace kt

class OdometryProvider {
  public:
    OdometryProvider() {}
    virtual ~OdometryProvider() {}

    virtual CloudSlice::Odometry getIncrementalTransformation(kt::Vector3f& trans, kt::Matrix3f& rot,
                                                              const DeviceArray2D<unsigned short>& depth,
                                                              const DeviceArray2D<PixelRGB>& image, uint64_t timestamp,
                                                              unsigned char* rgbImage, unsigned short* depthData) = 0;
    virtual void reset() = 0;

    // computeProjectiveMatrix (OdometryProvider.h:54-68): ksi = (t, rvec) -> row-major 4x4 [Rodrigues(rvec) | t]
    static void computeProjectiveMatrix(const double ksi[6], double Rt[16])
    {
        double R[9];
        ktSafeCall(kt_host_rodrigues(&ksi[3], R));
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Rt[i * 4 + j] = (i == j) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Rt[i * 4 + j] = R[i * 3 + j];
            Rt[i * 4 + 3] = ksi[i];
        }
    }
};
