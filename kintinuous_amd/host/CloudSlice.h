// CloudSlice.h -- one extracted slab of surface points plus the camera pose at extraction time: what the tracker hands
// to the CPU backend (frontend/CloudSlice.h:27-128), with the reference's constructor, fields and ownership.
// pcl::PointCloud<pcl::PointXYZRGB> is a std::vector of the 32-byte device point (same layout as pcl::PointXYZRGB, internal.h:156-184)
// and pcl::PointCloud<pcl::PointXYZRGBNormal> a std::vector of the 48-byte PointXYZRGBNormal (internal.h:186-229), unless PCL is on
// the include path -- then the PCL types are used (KT_WITH_PCL).
#pragma once

#include <stdint.h>
#include <cstring>
#include <vector>

#include "LinearAlgebra.h"
#include "PlaceRecognitionInput.h"
#include "Resolution.h"
#include "ThreadMutexObject.h"
#include "internal.h"

class CloudSlice {
  public:
    enum Dimension { XPlus, XMinus, YPlus, YMinus, ZPlus, ZMinus, FIRST, FINAL, TSDF };
    enum Odometry { ICP, GROUNDTRUTH, RGBD, FAIL };

    typedef std::vector<PointXYZRGB> PointCloud;
    typedef std::vector<PointXYZRGBNormal> PointCloudNormal;

    CloudSlice(PointCloud* cloud, Dimension dimension, Odometry odometry, const kt::Vector3f& cameraTranslation,
               const kt::Matrix3f& cameraRotation, uint64_t utime, uint64_t lagTime, unsigned char* rgbImage,
               unsigned char* tsdfImageColor = 0, unsigned char* tsdfImage = 0, unsigned short* depthData = 0,
               PlaceRecognitionInput* placeRecognitionFrame = 0)
        : cloud(cloud), processedCloud(0), dimension(dimension), odometry(odometry), cameraTranslation(cameraTranslation),
          cameraRotation(cameraRotation), poseIsam(false), utime(utime), lagTime(lagTime), tsdfImageColor(tsdfImageColor), tsdfImage(tsdfImage),
          depthData(0), placeRecognitionFrame(placeRecognitionFrame)
    {
        const int n = Resolution::get().numPixels();
        if (rgbImage != 0) {
            this->rgbImage = new unsigned char[n * 3];
            std::memcpy(this->rgbImage, rgbImage, (size_t)n * 3);
        } else {
            this->rgbImage = 0;
        }
        if (depthData != 0) {
            this->depthData = new unsigned short[n];
            std::memcpy(this->depthData, depthData, (size_t)n * 2);
        }
    }
    virtual ~CloudSlice()
    {
        delete cloud;
        delete processedCloud;
        delete[] rgbImage;
        delete[] tsdfImageColor;
        delete[] tsdfImage;
        delete[] depthData;
    }

    PointCloud* cloud;
    PointCloudNormal* processedCloud;
    Dimension dimension;
    Odometry odometry;
    kt::Vector3f cameraTranslation;
    kt::Matrix3f cameraRotation;
    ThreadMutexObject<bool> poseIsam;
    uint64_t utime;
    uint64_t lagTime;
    unsigned char* rgbImage;
    unsigned char* tsdfImageColor;
    unsigned char* tsdfImage;
    unsigned short* depthData;
    PlaceRecognitionInput* placeRecognitionFrame;

  private:
    CloudSlice();
    CloudSlice(const CloudSlice&);
    CloudSlice& operator=(const CloudSlice&);
};
