// GroundTruthOdometry.h -- "odometry" from a recorded camera trajectory (-p <file>), frontend/GroundTruthOdometry.cpp:25-112: the pose
// of a frame is the previous pose composed with the trajectory's increment between the two frames' time stamps, expressed in the
// volume's axes.  The float arithmetic (Eigen 3.2's evaluation order) lives in the library: kt_host_ground_truth_pose.
#pragma once

#include <array>
#include <functional>
#include <map>
#include <vector>

#include "OdometryProvider.h"

class GroundTruthOdometry : public OdometryProvider {
  public:
    // keyed like the reference's map: uint64_t keys compared as int (std::less<int>, KintinuousTracker.h:160-163)
    typedef std::map<uint64_t, std::array<float, 12>, std::less<int> > Trajectory;

    GroundTruthOdometry(std::vector<kt::Vector3f>& tvecs_, std::vector<kt::Matrix3f>& rmats_, Trajectory& camera_trajectory, uint64_t& last_utime)
        : tvecs_(tvecs_), rmats_(rmats_), camera_trajectory(camera_trajectory), last_utime(last_utime) {}

    // GroundTruthOdometry.cpp:42-74
    CloudSlice::Odometry getIncrementalTransformation(kt::Vector3f& trans, kt::Matrix3f& rot, const DeviceArray2D<unsigned short>&,
                                                      const DeviceArray2D<PixelRGB>&, uint64_t timestamp, unsigned char*, unsigned short*)
    {
        rot = rmats_.back();
        trans = tvecs_.back();
        if (last_utime != 0 && !camera_trajectory.empty()) {   // :50 -- a previous stamp of 0 reads as "no previous frame"
            const std::array<float, 12>& A = camera_trajectory[last_utime];   // operator[], as the reference
            const std::array<float, 12>& B = camera_trajectory[timestamp];
            kt_host_ground_truth_pose(A.data(), B.data(), rmats_.back().data(), tvecs_.back().data(), rot.data(), trans.data());
        }
        return CloudSlice::GROUNDTRUTH;
    }

    // :89-111: a frame whose stamp has no trajectory entry is dropped before anything else happens
    bool preRun(unsigned char*, unsigned short*, uint64_t timestamp) { return camera_trajectory.find(timestamp) != camera_trajectory.end(); }

    void reset() {}

  private:
    std::vector<kt::Vector3f>& tvecs_;
    std::vector<kt::Matrix3f>& rmats_;
    Trajectory& camera_trajectory;
    uint64_t& last_utime;
};
