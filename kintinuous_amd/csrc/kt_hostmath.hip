// kt_hostmath.hip -- host-side math of one Gauss-Newton step, exported through the C-ABI for callers that drive the
// per-operator entry points themselves (kintinuous_amd/host/ICPOdometry.h).  No GPU work; the same inline functions run
// in the epilogue of the reduction kernels (kt_track.hpp).  Replaces the Eigen / OpenCV calls of ICPOdometry.cpp:127-178.
#include "kt_track.hpp"

#include <math.h>
#include <string.h>

extern "C" {

int kt_host_ldlt_solve6(const double A[36], const double b[6], double x[6])
{
    KT_ARG(A && b && x);
    double tmp[36];
    for (int i = 0; i < 36; ++i) tmp[i] = A[i];
    kt_ldlt_solve6(tmp, b, x);
    return KT_OK;
}

int kt_host_rodrigues(const double r[3], double R[9])
{
    KT_ARG(r && R);
    kt_rodrigues(r, R);
    return KT_OK;
}

int kt_host_mat33_inverse(const float m[9], float out[9])
{
    KT_ARG(m && out);
    kt_mat33_inverse(m, out);
    return KT_OK;
}

int kt_host_pose_update(const double x[6], double resultRt[16], const float Rprev[9], const float tprev[3], float Rcurr[9], float tcurr[3])
{
    KT_ARG(x && resultRt && Rprev && tprev && Rcurr && tcurr);
    kt_pose_update(x, resultRt, Rprev, tprev, Rcurr, tcurr);
    return KT_OK;
}

}  // extern "C"

// KintinuousTracker::rodrigues2 (KintinuousTracker.cpp:1210-1255), the axis-angle of a rotation matrix in double as cv::Rodrigues'
// matrix branch computes it.  The reference's JacobiSVD re-orthonormalisation (R = U V^T) is not restated: on a rotation matrix the
// polar factor is the matrix up to float rounding.
static void axis_angle_of(const float R[9], float out[3])
{
    double ax = (double)(R[7] - R[5]), ay = (double)(R[2] - R[6]), az = (double)(R[3] - R[1]);
    const double sine = sqrt((ax * ax + ay * ay + az * az) * 0.25);
    double cosine = (double)((R[0] + R[4] + R[8]) - 1) * 0.5;
    cosine = cosine > 1. ? 1. : cosine < -1. ? -1. : cosine;
    double angle = acos(cosine);
    if (sine < 1e-5) {
        if (cosine > 0) ax = ay = az = 0;
        else {  // angle near pi: the axis comes from the diagonal
            double h = (R[0] + 1) * 0.5;
            ax = sqrt(h > 0.0 ? h : 0.0);
            h = (R[4] + 1) * 0.5;
            ay = sqrt(h > 0.0 ? h : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
            h = (R[8] + 1) * 0.5;
            az = sqrt(h > 0.0 ? h : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(ax) < fabs(ay) && fabs(ax) < fabs(az) && (R[5] > 0) != (ay * az > 0)) az = -az;
            angle /= sqrt(ax * ax + ay * ay + az * az);
            ax *= angle; ay *= angle; az *= angle;
        }
    } else {
        double k = 1 / (2 * sine);
        k *= angle;
        ax *= k; ay *= k; az *= k;
    }
    out[0] = (float)ax; out[1] = (float)ay; out[2] = (float)az;
}

extern "C" void kt_host_reposition_cube(const float R[9], const float tlast[3], float volume_size, const float voxel_size[3], int thresh,
                                        float basis[3])
{
    // KintinuousTracker::repositionCube :384-442: the cube's corner swings on a circle of half the cube size around the camera,
    // following the heading (rotation about y); it only moves when the camera would otherwise trip the shift threshold
    float aa[3];
    axis_angle_of(R, aa);
    const float heading = aa[1];
    const float PI = 3.14159265359f;
    const float radius = (float)(volume_size * 0.5);
    // global ::cos / ::sin on a float argument: the C library's double functions (the float sum is widened), then float * double
    const float moved[3] = {(float)(radius * (::cos((double)(heading + (PI / 2))) + 1.0f)), basis[1],
                            (float)(radius * (::sin((double)(heading - (PI / 2))) + 1.0f))};
    bool trips = false;
    for (int k = 0; k < 3; ++k) {
        // the clamped floor of :409-433
        const int f = (int)floorf((tlast[k] - moved[k]) / voxel_size[k]);
        const int v = f < 0 ? (-thresh > f ? -thresh : f) : (thresh < f ? thresh : f);
        trips = trips || v >= thresh || v <= -thresh;
    }
    if (trips) { basis[0] = moved[0]; basis[2] = moved[2]; }
}

// The movement measure of the place-recognition tap (KintinuousTracker.cpp:607-611): (|rodrigues2(Rcurr^-1 * Rlast)| + |c - clast|) / 2,
// every step in float as Eigen evaluates it (cofactor inverse, 3x3 product summed left to right, norms as sqrt of x^2 + y^2 + z^2).
extern "C" float kt_host_place_recognition_movement(const float Rcurr[9], const float cam[3], const float Rlast[9], const float camLast[3])
{
    float inv[9], rel[9], aa[3];
    kt_mat33_inverse(Rcurr, inv);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rel[i * 3 + j] = (inv[i * 3] * Rlast[j] + inv[i * 3 + 1] * Rlast[3 + j]) + inv[i * 3 + 2] * Rlast[6 + j];
    axis_angle_of(rel, aa);
    const float rnorm = sqrtf((aa[0] * aa[0] + aa[1] * aa[1]) + aa[2] * aa[2]);
    const float d[3] = {cam[0] - camLast[0], cam[1] - camLast[1], cam[2] - camLast[2]};
    const float tnorm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float alpha = 1.f;
    return (rnorm + alpha * tnorm) / 2;
}

// ---- host math of the other two odometry providers (for callers that drive the operators themselves) ----

extern "C" int kt_host_compute_krk(const double resultRt[16], double fx, double fy, double cx, double cy, float krkinv[9], float kt[3])
{
    KT_ARG(resultRt && krkinv && kt);
    kt_level_k k;
    k.fx = fx; k.fy = fy; k.cx = cx; k.cy = cy;
    kt_compute_krk(resultRt, k, krkinv, kt);
    return KT_OK;
}

// KintinuousTracker::loadTrajectory, KintinuousTracker.cpp:244-256: T.setIdentity(); T.pretranslate(t).rotate(q) -- the linear part is
// Quaternionf::toRotationMatrix() (no normalisation), the translation is t.  T = {R row-major (9), t (3)}.
extern "C" void kt_host_trajectory_pose(const float pose7[7], float T[12])
{
    const float* p = pose7;
    const float qx = p[3], qy = p[4], qz = p[5], qw = p[6];
    const float x2 = 2.f * qx, y2 = 2.f * qy, z2 = 2.f * qz;
    const float wx = x2 * qw, wy = y2 * qw, wz = z2 * qw;
    const float xx = x2 * qx, xy = y2 * qx, xz = z2 * qx;
    const float yy = y2 * qy, yz = z2 * qy, zz = z2 * qz;
    const float out[12] = {1.f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.f - (xx + yy), p[0], p[1], p[2]};
    memcpy(T, out, sizeof(out));
}

// GroundTruthOdometry::getIncrementalTransformation, GroundTruthOdometry.cpp:42-74, with Eigen 3.2's float evaluation order:
// delta = Ta^-1 * Tb as Isometry3f (3x3 products, translation = linear * t + t), then currentTsdf * M^-1 * delta * M left to right as
// 4x4 products.  M permutes and negates axes, so the first and last product are exact column moves; the middle one rounds, with
// its sums running over the moved columns.  A, B: trajectory poses of the previous and the current stamp (kt_host_trajectory_pose).
extern "C" void kt_host_ground_truth_pose(const float A[12], const float B[12], const float Rlast[9], const float tlast[3], float Rcurr[9],
                                          float tcurr[3])
{
    float Ainv[9], ainv[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Ainv[i * 3 + j] = A[j * 3 + i];
    for (int i = 0; i < 3; ++i) ainv[i] = ((-Ainv[i * 3]) * A[9] + (-Ainv[i * 3 + 1]) * A[10]) + (-Ainv[i * 3 + 2]) * A[11];
    float delta[4][4];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) delta[i][j] = (Ainv[i * 3] * B[j] + Ainv[i * 3 + 1] * B[3 + j]) + Ainv[i * 3 + 2] * B[6 + j];
        delta[i][3] = ((Ainv[i * 3] * B[9] + Ainv[i * 3 + 1] * B[10]) + Ainv[i * 3 + 2] * B[11]) + ainv[i];
    }
    delta[3][0] = delta[3][1] = delta[3][2] = 0.f;
    delta[3][3] = 1.f;
    float Rn[9], tn[3];
    for (int i = 0; i < 3; ++i) {
        // row i of currentTsdf * M^-1: (z column, -x column, -y column, translation)
        const float moved[4] = {Rlast[i * 3 + 2], -Rlast[i * 3], -Rlast[i * 3 + 1], tlast[i]};
        float q[4];
        for (int j = 0; j < 4; ++j) q[j] = ((moved[0] * delta[0][j] + moved[1] * delta[1][j]) + moved[2] * delta[2][j]) + moved[3] * delta[3][j];
        // ... * M: columns (-q1, -q2, q0, q3)
        Rn[i * 3] = -q[1];
        Rn[i * 3 + 1] = -q[2];
        Rn[i * 3 + 2] = q[0];
        tn[i] = q[3];
    }
    memcpy(Rcurr, Rn, sizeof(Rn));
    memcpy(tcurr, tn, sizeof(tn));
}

