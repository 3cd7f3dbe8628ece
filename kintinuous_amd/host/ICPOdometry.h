// ICPOdometry.h -- projective point-to-plane ICP against the predicted model maps (frontend/ICPOdometry.cpp:68-186): one call of
// kt_icp_track -- all 19 Gauss-Newton iterations enqueued back to back, the 6x6 LDLT solve in double, Rodrigues and the SE(3)
// composition in each reduction kernel's epilogue, pose in / pose out.  (The reference iterates on the host: icpStep, a blocking copy of
// 29 sums and an Eigen solve per iteration; that loop can still be composed from kt_icp_step + kt_host_ldlt_solve6 +
// kt_host_pose_update, and gives bit-identical poses: tests/test_gpu_track.py::test_icp_track_matches_the_stepwise_loop.)
// This is what KintinuousTracker's operator path (-ops) tracks with; its default path runs the same chain inside
// kt_tracker_process_frame.
#pragma once

#include <cmath>
#include <vector>

#include "OdometryProvider.h"
#include "internal.h"

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
        kt::Matrix3f Rcurr = Rprev;
        kt::Vector3f tcurr = tprev;
        const float *vc[LEVELS], *nc[LEVELS], *vp[LEVELS], *np[LEVELS];
        for (int l = 0; l < LEVELS; ++l) {
            vc[l] = vmaps_curr_[l].ptr(); nc[l] = nmaps_curr_[l].ptr();
            vp[l] = vmaps_g_prev_[l].ptr(); np[l] = nmaps_g_prev_[l].ptr();
        }
        float A_last[36], residual[2];
        ktSafeCall(kt_icp_track(::kt::device::context(), vc, nc, vp, np, vmaps_curr_[0].cols(), vmaps_curr_[0].rows() / 3, kt::abi(intr), kt::abi(kt::dev(Rprev)),
                                kt::abi(kt::dev(tprev)), icp_iterations_, distThres_, angleThres_, reinterpret_cast<kt_mat33*>(Rcurr.data()), tcurr.data(),
                                A_last, residual));
        for (int i = 0; i < 36; ++i) lastA[i] = (double)A_last[i];
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
    float distThres_, angleThres_;
};
