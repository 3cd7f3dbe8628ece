// kt_hostmath.hip -- host-side math of one Gauss-Newton step, exported through the C-ABI for callers that drive the
// per-operator entry points themselves (kintinuous_amd/host/ICPOdometry.h).  No GPU work; the same inline functions run
// in the epilogue of the reduction kernels (kt_track.hpp).  Replaces the Eigen / OpenCV calls of ICPOdometry.cpp:127-178.
#include "kt_track.hpp"

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
