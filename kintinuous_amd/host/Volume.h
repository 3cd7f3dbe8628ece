// Volume.h -- edge length (metres) and resolution (voxels per edge) of the TSDF cube, process-wide (interface of frontend/Volume.h:27-57).
// The reference fixes the resolution at compile time (VOLUME_X/Y/Z, internal.h:243); here it is the second argument of the first get().
#pragma once

#include <cstdio>
#include <cstdlib>

#include "internal.h"

class Volume {
    float edge;      // metres
    int voxels;      // per edge
    float3 voxel;    // metres per voxel, per axis

    Volume(float edge_m, int n) : edge(edge_m), voxels(n)
    {
        if (!(edge_m > 0) || n <= 0) {
            std::fprintf(stderr, "Volume: get(size, resolution) must be called with the cube's size before anything asks for it\n");
            std::abort();
        }
        const float v = edge_m / (float)n;
        voxel = make_float3(v, v, v);
    }

  public:
    static Volume& get(float volumeSize = 0, int resolution = 0)
    {
        static Volume the_one(volumeSize, resolution);
        return the_one;
    }
    int getResolution() const { return voxels; }
    const float& getVolumeSize() { return edge; }
    const float3& getVoxelSizeMeters() { return voxel; }
};

namespace kt {
inline int volSide() { return Volume::get().getResolution(); }
}
