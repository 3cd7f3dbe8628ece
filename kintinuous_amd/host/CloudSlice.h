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

// deep copy of a frame buffer that the slice owns from then on (null stays null)
template <class T> inline T* ktCloneFrame(const T* src, size_t count)
{
    if (!src) return 0;
    T* copy = new T[count];
    std::memcpy(copy, src, count * sizeof(T));
    return copy;
}

class CloudSlice {
    CloudSlice();                               // slices are made from an extracted cloud only, and never copied:
    CloudSlice(const CloudSlice&);              // they own their buffers
    CloudSlice& operator=(const CloudSlice&);

  public:
    // the face of the cube a slice left through (or what kind of snapshot it is) and the odometry that produced its pose
    enum Dimension { XPlus = 0, XMinus = 1, YPlus = 2, YMinus = 3, ZPlus = 4, ZMinus = 5, FIRST = 6, FINAL = 7, TSDF = 8 };
    enum Odometry { ICP = 0, GROUNDTRUTH = 1, RGBD = 2, FAIL = 3 };

    typedef std::vector<PointXYZRGB> PointCloud;
    typedef std::vector<PointXYZRGBNormal> PointCloudNormal;

    // ---- payload (public, as the backend reads and fills it) ----
    PointCloud* cloud;                          // extracted points; owned
    PointCloudNormal* processedCloud;           // filled by CloudSliceProcessor; owned
    Dimension dimension;
    Odometry odometry;
    kt::Vector3f cameraTranslation;
    kt::Matrix3f cameraRotation;
    ThreadMutexObject<bool> poseIsam;           // set once iSAM has optimised the slice's pose
    uint64_t utime, lagTime;
    unsigned char* rgbImage;                    // copies of the frame the slice was cut at; owned
    unsigned char *tsdfImageColor, *tsdfImage;  // live-view renderings (TSDF slices); owned, taken over from the caller
    unsigned short* depthData;
    PlaceRecognitionInput* placeRecognitionFrame;

    // KintinuousTracker.cpp:1186: cloud and the two tsdf images are taken over, rgbImage / depthData are copied (one frame each)
    CloudSlice(PointCloud* cloud_, Dimension dimension_, Odometry odometry_, const kt::Vector3f& cameraTranslation_,
               const kt::Matrix3f& cameraRotation_, uint64_t utime_, uint64_t lagTime_, unsigned char* rgbImage_,
               unsigned char* tsdfImageColor_ = 0, unsigned char* tsdfImage_ = 0, unsigned short* depthData_ = 0,
               PlaceRecognitionInput* placeRecognitionFrame_ = 0)
        : cloud(cloud_), processedCloud(0), dimension(dimension_), odometry(odometry_), cameraTranslation(cameraTranslation_),
          cameraRotation(cameraRotation_), poseIsam(false), utime(utime_), lagTime(lagTime_),
          rgbImage(ktCloneFrame(rgbImage_, (size_t)Resolution::get().numPixels() * 3)), tsdfImageColor(tsdfImageColor_), tsdfImage(tsdfImage_),
          depthData(ktCloneFrame(depthData_, (size_t)Resolution::get().numPixels())), placeRecognitionFrame(placeRecognitionFrame_)
    {
    }

    virtual ~CloudSlice()
    {
        delete[] depthData;
        delete[] tsdfImage;
        delete[] tsdfImageColor;
        delete[] rgbImage;
        delete processedCloud;
        delete cloud;
    }
};
