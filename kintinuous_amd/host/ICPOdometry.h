// ICPOdometry.h -- projective point-to-plane ICP against the predicted model maps, driven from the host one iteration
// at a time exactly like the reference (frontend/ICPOdometry.cpp:68-186): icpStep -> 6x6 LDLT solve in double ->
// Rodrigues -> SE(3) composition.  This is the per-operator path over kt_icp_step (one host sync per iteration);
// KintinuousTracker's default path runs the same iterations device-resident (kt_tracker_process_frame).  Both produce
// bit-identical poses (tests/test_gpu_host_shell.py).
#pragma once

#include <cmath>
#include <vector>

#include "OdometryProvider.h"

class ICPOdometry : public OdometryProvider {
  public:
    static const int LEVELS = 4;  // ICPOdometry.h:52

    ICPOdometry(std::vector<kt::Vector3f>& tvecs_, std::vector<kt::Matrix3f>& rmats_, std::vector<DeviceArray2D<float> >& vmaps_g_prev_,
                std::vector<DeviceArray2D<float> >& nmaps_g_prev_, std::vector<DeviceArray2D<float> >& vmaps_curr_,
                std::vector<DeviceArray2D<float> >& nmaps_curr_, Intr& intr, bool fastOdometry = false, float distThresh = 0.10f,
                float angleThresh = std::sin(20.f * 3.14159254f / 180.f))
        : tvecs_(tvecs_), rmats_(rmats_), vmaps_g_prev_(vmaps_g_prev_), nmaps_g_prev_(nmaps_g_prev_), vmaps_curr_(vmaps_curr_),
          nmaps_curr_(nmaps_curr_), intr(intr), distThres_(distThresh), angleThres_(angleThresh)
    {
        // ICPOdometry.cpp:42-55
        const int normal[LEVELS] = {10, 5, 4, 0}, fast[LEVELS] = {0, 10, 5, 0};
        for (int i = 0; i < LEVELS; ++i) icp_iterations_[i] = fastOdometry ? fast[i] : normal[i];
        for (int i = 0; i < 36; ++i) lastA[i] = 0;
    }

    CloudSlice::Odometry getIncrementalTransformation(kt::Vector3f& trans, kt::Matrix3f& rot, const DeviceArray2D<unsigned short>&,
                                                      const DeviceArray2D<PixelRGB>&, uint64_t, unsigned char*, unsigned short*)
    {
        const kt::Matrix3f Rprev = rmats_.back();
        const kt::Vector3f tprev = tvecs_.back();
        kt::Matrix3f Rcurr = Rprev, Rprev_inv;
        kt::Vector3f tcurr = tprev;
        ktSafeCall(kt_host_mat33_inverse(Rprev.data(), Rprev_inv.data()));
        double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

        for (int level_index = LEVELS - 1; level_index >= 0; --level_index) {
            for (int iter = 0; iter < icp_iterations_[level_index]; ++iter) {
                float A_icp[36], b_icp[6], residual[2];
                icpStep(kt::dev(Rcurr), kt::dev(tcurr), vmaps_curr_[level_index],
                        nmaps_curr_[level_index], kt::dev(Rprev_inv), kt::dev(tprev),
                        intr(level_index), vmaps_g_prev_[level_index], nmaps_g_prev_[level_index], distThres_, angleThres_, sumDataSE3,
                        outDataSE3, A_icp, b_icp, residual, 128, 64);
                double dA[36], db[6], result[6];
                for (int i = 0; i < 36; ++i) lastA[i] = dA[i] = (double)A_icp[i];
                for (int i = 0; i < 6; ++i) db[i] = (double)b_icp[i];
                ktSafeCall(kt_host_ldlt_solve6(dA, db, result));
                // resultRt = currRt * resultRt;  currentT = [Rprev | tprev] * resultRt^-1   (ICPOdometry.cpp:133-178)
                ktSafeCall(kt_host_pose_update(result, resultRt, Rprev.data(), tprev.data(), Rcurr.data(), tcurr.data()));
            }
        }
        trans = tcurr;
        rot = Rcurr;
        return CloudSlice::ICP;
    }

    const double* getLastA() const { return lastA; }  // the reference exposes its LU inverse as getCovariance()
    void reset() {}

  private:
    int icp_iterations_[LEVELS];
    std::vector<kt::Vector3f>& tvecs_;
    std::vector<kt::Matrix3f>& rmats_;
    std::vector<DeviceArray2D<float> >& vmaps_g_prev_;
    std::vector<DeviceArray2D<float> >& nmaps_g_prev_;
    std::vector<DeviceArray2D<float> >& vmaps_curr_;
    std::vector<DeviceArray2D<float> >& nmaps_curr_;
    Intr& intr;
    double lastA[36];
    DeviceArray<JtJJtrSE3> sumDataSE3, outDataSE3;
    float distThres_, angleThres_;
};
