// Volume.h -- volume size singleton (frontend/Volume.h:27-57).  The reference's compile-time VOLUME_X/Y/Z become the
// runtime resolution N given on the first call.
#pragma once
#include <cassert>
#include "internal.h"

class Volume {
  public:
    static Volume& get(float volumeSize = 0, int resolution = 0)
    {
        static Volume instance(volumeSize, resolution);
        return instance;
    }
    const float& getVolumeSize() { return volumeSize; }
    const float3& getVoxelSizeMeters() { return voxelSizeMeters; }
    int getResolution() const { return resolution; }

  private:
    Volume(float inVolumeSize, int inResolution) : volumeSize(inVolumeSize), resolution(inResolution)
    {
        assert(volumeSize > 0 && resolution > 0);
        voxelSizeMeters.x = voxelSizeMeters.y = voxelSizeMeters.z = volumeSize / float(resolution);
    }
    const float volumeSize;
    const int resolution;
    float3 voxelSizeMeters;
};

namespace kt {
inline int volSide() { return Volume::get().getResolution(); }
}
