// RGBDOdometry.h -- dense photometric (RGB-D) odometry, alone (-r) or combined with ICP (-ri), driven from the host one operator at a
// time exactly like the reference (frontend/RGBDOdometry.cpp:33-393): populateRGBDData -> derivative images -> per level
// projectToPointCloud, per iteration computeRgbResidual [-> icpStep] -> rgbStep -> 6x6 LDLT in double -> SE(3) composition.
// This is the per-operator path (two or three host syncs per iteration); KintinuousTracker's default path runs the same iterations
// device-resident.  Both produce identical poses (tests/test_gpu_host_shell.py).
#pragma once

#include <cmath>
#include <utility>
#include <vector>

#include "ConfigArgs.h"
#include "OdometryProvider.h"
#include "Resolution.h"

class RGBDOdometry : public OdometryProvider {
  public:
    static const int NUM_PYRS = 4;  // RGBDOdometry.h:97

    RGBDOdometry(std::vector<kt::Vector3f>& tvecs_, std::vector<kt::Matrix3f>& rmats_, std::vector<DeviceArray2D<float> >& vmaps_g_prev_,
                 std::vector<DeviceArray2D<float> >& nmaps_g_prev_, std::vector<DeviceArray2D<float> >& vmaps_curr_,
                 std::vector<DeviceArray2D<float> >& nmaps_curr_, Intr& intr, float distThresh = 0.10f,
                 float angleThresh = std::sin(20.f * 3.14159254f / 180.f))
        : tvecs_(tvecs_), rmats_(rmats_), vmaps_g_prev_(vmaps_g_prev_), nmaps_g_prev_(nmaps_g_prev_), vmaps_curr_(vmaps_curr_),
          nmaps_curr_(nmaps_curr_), intr(intr), SOBEL_SIZE(3), SOBEL_SCALE(1.0 / std::pow(2.0, SOBEL_SIZE)), MAX_DEPTH_DELTA(0.07),
          MAX_DEPTH(6.0), intrinsics(intr.fx, intr.fy, intr.cx, intr.cy), distThres_(distThresh), angleThres_(angleThresh)
    {
        // RGBDOdometry.cpp:77-112
        const ConfigArgs& args = ConfigArgs::get();
        const int rgbd[NUM_PYRS] = {10, 7, 7, 7}, rgbdFast[NUM_PYRS] = {0, 10, 7, 0}, joint[NUM_PYRS] = {10, 5, 4, 0};
        for (int i = 0; i < NUM_PYRS; ++i)
            iterations[i] = args.fastOdometry ? rgbdFast[i] : (args.useRGBDICP ? joint[i] : rgbd[i]);
        const float grad[NUM_PYRS] = {12, 5, 3, 1};
        for (int i = 0; i < NUM_PYRS; ++i) minimumGradientMagnitudes[i] = grad[i];
        for (int i = 0; i < 36; ++i) lastA[i] = 0;
        // the pyramids, RGBDOdometry.cpp:44-68 (DeviceArray2D::create(rows, cols))
        for (int i = 0; i < NUM_PYRS; ++i) {
            const int rows = Resolution::get().rows() >> i, cols = Resolution::get().cols() >> i;
            lastDepth[i].create(rows, cols); lastImage[i].create(rows, cols);
            nextDepth[i].create(rows, cols); nextImage[i].create(rows, cols);
            nextdIdx[i].create(rows, cols); nextdIdy[i].create(rows, cols);
            pointClouds[i].create(rows, cols);
            corresImg[i].create(rows, cols);
        }
    }

    // RGBDOdometry.cpp:160-163
    void firstRun(const DeviceArray2D<unsigned short>& depth, const DeviceArray2D<PixelRGB>& image) { populateRGBDData(depth, image, lastDepth, lastImage); }

    CloudSlice::Odometry getIncrementalTransformation(kt::Vector3f& trans, kt::Matrix3f& rot, const DeviceArray2D<unsigned short>& depth,
                                                      const DeviceArray2D<PixelRGB>& image, uint64_t, unsigned char*, unsigned short*)
    {
        const bool joint = ConfigArgs::get().useRGBDICP;
        const kt::Matrix3f Rprev = rmats_.back();
        const kt::Vector3f tprev = tvecs_.back();
        kt::Matrix3f Rcurr = Rprev, Rprev_inv;
        kt::Vector3f tcurr = tprev;
        ktSafeCall(kt_host_mat33_inverse(Rprev.data(), Rprev_inv.data()));

        populateRGBDData(depth, image, nextDepth, nextImage);
        for (int i = 0; i < NUM_PYRS; ++i) computeDerivativeImages(nextImage[i], nextdIdx[i], nextdIdy[i]);

        double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int i = NUM_PYRS - 1; i >= 0; --i) {
            projectToPointCloud(lastDepth[i], pointClouds[i], intrinsics, i);
            const double div = (double)(1 << i);   // IntrDoublePrecision::operator(), internal.h:268-272
            const double lfx = intrinsics.fx / div, lfy = intrinsics.fy / div, lcx = intrinsics.cx / div, lcy = intrinsics.cy / div;
            for (int j = 0; j < iterations[i]; ++j) {
                // K R K^-1, K t of resultRt^-1 :213-231
                Mat33 krkInv;
                float3 kt_;
                ktSafeCall(kt_host_compute_krk(resultRt, lfx, lfy, lcx, lcy, &krkInv.data[0].x, &kt_.x));
                int sigma = 0, rgbSize = 0;
                computeRgbResidual((float)(std::pow((double)minimumGradientMagnitudes[i], 2.0) / std::pow(SOBEL_SCALE, 2.0)), nextdIdx[i], nextdIdy[i],
                                   lastDepth[i], nextDepth[i], lastImage[i], nextImage[i], corresImg[i], sumResidualRGB, (float)MAX_DEPTH_DELTA, kt_,
                                   krkInv, sigma, rgbSize, 128, 256);
                // :253, as written: sqrt(count) unless sigma / count == 0
                const float sigmaVal = std::sqrt(((float)sigma / rgbSize == 0) ? 1 : rgbSize);

                float A_icp[36], b_icp[6], residual[2];
                if (joint)
                    icpStep(kt::dev(Rcurr), kt::dev(tcurr), vmaps_curr_[i], nmaps_curr_[i], kt::dev(Rprev_inv), kt::dev(tprev), intr(i),
                            vmaps_g_prev_[i], nmaps_g_prev_[i], distThres_, angleThres_, sumDataSE3, outDataSE3, A_icp, b_icp, residual, 128, 64);
                float A_rgbd[36], b_rgbd[6];
                rgbStep(corresImg[i], sigmaVal, pointClouds[i], intr(i).fx, intr(i).fy, nextdIdx[i], nextdIdy[i], (float)SOBEL_SCALE, sumDataSE3,
                        outDataSE3, A_rgbd, b_rgbd, 128, 64);

                double dA[36], db[6], result[6];
                if (joint) {   // :316-321
                    const double w = 10;
                    for (int k = 0; k < 36; ++k) dA[k] = (double)A_rgbd[k] + w * w * (double)A_icp[k];
                    for (int k = 0; k < 6; ++k) db[k] = (double)b_rgbd[k] + w * (double)b_icp[k];
                } else {
                    for (int k = 0; k < 36; ++k) dA[k] = (double)A_rgbd[k];
                    for (int k = 0; k < 6; ++k) db[k] = (double)b_rgbd[k];
                }
                for (int k = 0; k < 36; ++k) lastA[k] = dA[k];
                ktSafeCall(kt_host_ldlt_solve6(dA, db, result));
                // resultRt = currRt * resultRt;  currentT = [Rprev | tprev] * resultRt^-1   :339-374
                ktSafeCall(kt_host_pose_update(result, resultRt, Rprev.data(), tprev.data(), Rcurr.data(), tcurr.data()));
            }
        }
        for (int i = 0; i < NUM_PYRS; ++i) {
            std::swap(lastDepth[i], nextDepth[i]);
            std::swap(lastImage[i], nextImage[i]);
        }
        // :383-387: an increment of more than 0.3 m is discarded (the norm is a float, the comparison is in double)
        const float d0 = tcurr(0) - tprev(0), d1 = tcurr(1) - tprev(1), d2 = tcurr(2) - tprev(2);
        if ((double)std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) > 0.3) {
            Rcurr = Rprev;
            tcurr = tprev;
        }
        trans = tcurr;
        rot = Rcurr;
        return CloudSlice::RGBD;
    }

    const double* getLastA() const { return lastA; }  // the reference exposes its LU inverse as getCovariance()
    void reset() {}

  private:
    // RGBDOdometry.cpp:140-158
    void populateRGBDData(const DeviceArray2D<unsigned short>& depth, const DeviceArray2D<PixelRGB>& image, DeviceArray2D<float>* destDepths,
                          DeviceArray2D<unsigned char>* destImages)
    {
        shortDepthToMetres(depth, destDepths[0], (int)(MAX_DEPTH * 1000));
        for (int i = 0; i + 1 < NUM_PYRS; ++i) pyrDownGaussF(destDepths[i], destDepths[i + 1]);
        imageBGRToIntensity(image, destImages[0]);
        for (int i = 0; i + 1 < NUM_PYRS; ++i) pyrDownUcharGauss(destImages[i], destImages[i + 1]);
    }

    std::vector<kt::Vector3f>& tvecs_;
    std::vector<kt::Matrix3f>& rmats_;
    std::vector<DeviceArray2D<float> >& vmaps_g_prev_;
    std::vector<DeviceArray2D<float> >& nmaps_g_prev_;
    std::vector<DeviceArray2D<float> >& vmaps_curr_;
    std::vector<DeviceArray2D<float> >& nmaps_curr_;
    Intr& intr;
    DeviceArray<JtJJtrSE3> sumDataSE3, outDataSE3;
    DeviceArray<int2> sumResidualRGB;
    const int SOBEL_SIZE;
    const double SOBEL_SCALE, MAX_DEPTH_DELTA, MAX_DEPTH;
    DeviceArray2D<float> lastDepth[NUM_PYRS], nextDepth[NUM_PYRS];
    DeviceArray2D<unsigned char> lastImage[NUM_PYRS], nextImage[NUM_PYRS];
    DeviceArray2D<short> nextdIdx[NUM_PYRS], nextdIdy[NUM_PYRS];
    DeviceArray2D<DataTerm> corresImg[NUM_PYRS];
    DeviceArray2D<float3> pointClouds[NUM_PYRS];
    IntrDoublePrecision intrinsics;
    int iterations[NUM_PYRS];
    float minimumGradientMagnitudes[NUM_PYRS];
    double lastA[36];
    float distThres_, angleThres_;
};
