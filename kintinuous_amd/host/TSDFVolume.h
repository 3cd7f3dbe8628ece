// TSDFVolume.h / ColorVolume -- device volume containers with the reference's public surface (frontend/TSDFVolume.h:39-158,
// TSDFVolume.cpp:42-230; frontend/ColorVolume.h:36-94, ColorVolume.cpp).  Storage is DeviceArray2D<T>(N*N, N).
#pragma once

#include <algorithm>
#include <vector>

#include "LinearAlgebra.h"
#include "internal.h"

class TsdfVolume {
  public:
    enum { DEFAULT_CLOUD_BUFFER_SIZE = 10 * 1000 * 1000 };  // TSDFVolume.h:46

    explicit TsdfVolume(int resolution) : resolution_(resolution)
    {
        volume_.create(resolution * resolution, resolution);
        setSize(kt::Vector3f(3.0f, 3.0f, 3.0f));        // TSDFVolume.cpp:59-63 defaults
        setTsdfTruncDist(0.03f);
        reset();
    }
    void setSize(const kt::Vector3f& size) { size_ = size; setTsdfTruncDist(tranc_dist_); }
    // TSDFVolume.cpp:78-88: clamped below at 2.1 * the largest voxel edge
    void setTsdfTruncDist(float distance)
    {
        const float cx = size_(0) / resolution_, cy = size_(1) / resolution_, cz = size_(2) / resolution_;
        tranc_dist_ = std::max(distance, 2.1f * std::max(cx, std::max(cy, cz)));
    }
    DeviceArray2D<short>& data() { return volume_; }
    const DeviceArray2D<short>& data() const { return volume_; }
    const kt::Vector3f& getSize() const { return size_; }
    int getResolution() const { return resolution_; }
    kt::Vector3f getVoxelSize() const { return kt::Vector3f(size_(0) / resolution_, size_(1) / resolution_, size_(2) / resolution_); }
    float getTsdfTruncDist() const { return tranc_dist_; }
    void reset() { initVolume(volume_); }

    // TSDFVolume.cpp:135-172: returns a view of cloud_buffer holding the extracted points
    DeviceArray<PointXYZRGB> fetchCloud(DeviceArray<PointXYZRGB>& cloud_buffer, int3& voxelWrap, PtrStep<uchar4> color_volume,
                                        int minX, int maxX, int minY, int maxY, int minZ, int maxZ, int3 realVoxelWrap,
                                        int subsample = 1) const
    {
        if (cloud_buffer.empty()) cloud_buffer.create(DEFAULT_CLOUD_BUFFER_SIZE);
        const float3 device_volume_size = make_float3(size_(0), size_(1), size_(2));
        const size_t size = extractCloudSlice(volume_, device_volume_size, cloud_buffer, voxelWrap, color_volume, minX, maxX, minY,
                                              maxY, minZ, maxZ, subsample, realVoxelWrap);
        return DeviceArray<PointXYZRGB>(cloud_buffer.ptr(), size);
    }
    // TSDFVolume.cpp:174-190: tsdf as float in [-1, 1]
    void downloadTsdf(std::vector<float>& tsdf) const
    {
        std::vector<short> raw;
        int cols;
        volume_.download(raw, cols);
        tsdf.resize(raw.size());
        for (size_t i = 0; i < raw.size(); ++i) tsdf[i] = (float)raw[i] / 32767.0f;
    }

  private:
    int resolution_;
    kt::Vector3f size_;
    float tranc_dist_ = 0.03f;
    DeviceArray2D<short> volume_;
};

class ColorVolume {
  public:
    explicit ColorVolume(const TsdfVolume& tsdf) : resolution_(tsdf.getResolution())
    {
        color_volume_.create(resolution_ * resolution_, resolution_);
        reset();
    }
    void reset() { initColorVolume(color_volume_); }
    // ColorVolume.cpp:77-83: the volume is a DeviceArray2D<int> handed out BY VALUE (a shared handle); the operators take it as
    // PtrStep<uchar4> through DeviceArray2D's converting operator
    DeviceArray2D<int> data() const { return color_volume_; }

  private:
    int resolution_;
    DeviceArray2D<int> color_volume_;
};
