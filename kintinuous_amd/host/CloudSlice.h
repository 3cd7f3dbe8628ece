// CloudSlice.h -- one extracted slab of surface points plus the camera pose at extraction time: what the tracker hands
// to the CPU backend (frontend/CloudSlice.h:27-128).  pcl::PointCloud<pcl::PointXYZRGB> is replaced by a vector of the
// 32-byte device point (same layout as pcl::PointXYZRGB), so a maintainer can memcpy it into a PCL cloud.
#pragma once

#include <stdint.h>
#include <cstring>
#include <vector>

#include "LinearAlgebra.h"
#include "Resolution.h"
#include "internal.h"

class CloudSlice {
  public:
    enum Dimension { XPlus, XMinus, YPlus, YMinus, ZPlus, ZMinus, FIRST, FINAL, TSDF };
    enum Odometry { ICP, GROUNDTRUTH, RGBD, FAIL };

    typedef std::vector<PointXYZRGB> PointCloud;

    CloudSlice(PointCloud* cloud, Dimension dimension, Odometry odometry, const kt::Vector3f& cameraTranslation,
               const kt::Matrix3f& cameraRotation, uint64_t utime, uint64_t lagTime, const unsigned char* rgbImage = 0,
               const unsigned short* depthData = 0)
        : cloud(cloud), dimension(dimension), odometry(odometry), cameraTranslation(cameraTranslation),
          cameraRotation(cameraRotation), utime(utime), lagTime(lagTime), rgbImage(0), depthData(0)
    {
        const int n = Resolution::get().numPixels();
        if (rgbImage) {
            this->rgbImage = new unsigned char[n * 3];
            std::memcpy(this->rgbImage, rgbImage, (size_t)n * 3);
        }
        if (depthData) {
            this->depthData = new unsigned short[n];
            std::memcpy(this->depthData, depthData, (size_t)n * 2);
        }
    }
    virtual ~CloudSlice()
    {
        delete cloud;
        delete[] rgbImage;
        delete[] depthData;
    }

    PointCloud* cloud;
    Dimension dimension;
    Odometry odometry;
    kt::Vector3f cameraTranslation;
    kt::Matrix3f cameraRotation;
    uint64_t utime, lagTime;
    unsigned char* rgbImage;
    unsigned short* depthData;

  private:
    CloudSlice(const CloudSlice&);
    CloudSlice& operator=(const CloudSlice&);
};
