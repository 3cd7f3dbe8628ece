// LinearAlgebra.h -- the few Eigen types the reference's frontend API exposes (Vector3f, row-major Matrix3f, Matrix4f,
// Quaternionf(Matrix3f)), as plain structs: this image has no Eigen.  Only storage and the operations the shell needs.
#pragma once

#include <cmath>

namespace kt {

struct Vector3f {
    float v[3];
    Vector3f() : v{0, 0, 0} {}
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    float& operator()(int i) { return v[i]; }
    const float& operator()(int i) const { return v[i]; }
    float* data() { return v; }
    const float* data() const { return v; }
};

struct Vector3i { int v[3]; int& operator()(int i) { return v[i]; } const int& operator()(int i) const { return v[i]; } };

// Eigen::Matrix<float, 3, 3, Eigen::RowMajor>
struct Matrix3f {
    float m[9];
    Matrix3f() : m{1, 0, 0, 0, 1, 0, 0, 0, 1} {}
    float& operator()(int i, int j) { return m[i * 3 + j]; }
    const float& operator()(int i, int j) const { return m[i * 3 + j]; }
    float* data() { return m; }
    const float* data() const { return m; }
};

// Eigen::Matrix4f, exposed row-major here (DensePose::pose)
struct Matrix4f {
    float m[16];
    Matrix4f() : m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1} {}
    float& operator()(int i, int j) { return m[i * 4 + j]; }
    const float& operator()(int i, int j) const { return m[i * 4 + j]; }
};

// Eigen::Quaternionf(Matrix3f) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>)
struct Quaternionf {
    float x, y, z, w;
    explicit Quaternionf(const Matrix3f& R)
    {
        float t = R(0, 0) + R(1, 1) + R(2, 2);
        if (t > 0.0f) {
            t = std::sqrt(t + 1.0f);
            w = 0.5f * t;
            t = 0.5f / t;
            x = (R(2, 1) - R(1, 2)) * t;
            y = (R(0, 2) - R(2, 0)) * t;
            z = (R(1, 0) - R(0, 1)) * t;
        } else {
            int i = 0;
            if (R(1, 1) > R(0, 0)) i = 1;
            if (R(2, 2) > R(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0f);
            float q[3];
            q[i] = 0.5f * t;
            t = 0.5f / t;
            w = (R(k, j) - R(j, k)) * t;
            q[j] = (R(j, i) + R(i, j)) * t;
            q[k] = (R(k, i) + R(i, k)) * t;
            x = q[0]; y = q[1]; z = q[2];
        }
    }
};

}  // namespace kt
