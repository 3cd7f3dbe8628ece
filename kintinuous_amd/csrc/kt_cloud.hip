// kt_cloud.hip -- what CloudSliceProcessor::save does with the processed slices once a run has ended
// (backend/CloudSliceProcessor.cpp:180-231): one more pcl::VoxelGrid over the concatenated pcl::PointXYZRGBNormal cloud (when
// overlapping slices were extracted and are not to be kept) and pcl::io::savePCDFile(file, cloud, true).  Host code: the reference runs
// it once per run on the CPU, after the last frame; nothing here is on the frame path.  PCL 1.7 is not vendored with the reference
// (README.md:14-31), so this restates filters/impl/voxel_grid.hpp (applyFilter with downsample_all_data_) and io/impl/pcd_io.hpp
// (generateHeader + writeBinary) for that one point type; the oracle carries an independent restatement (kto_voxel_grid_normal,
// kto_pcd_binary) and tests/test_pcd.py holds the two against each other and against a numpy model.
#include "kt_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

// the 11 floats VoxelGrid averages per leaf: the 8 registered fields in POINT_CLOUD_REGISTER_POINT_STRUCT order
// (x y z rgb normal_x normal_y normal_z curvature; `rgb` as the float its bits spell, overwritten afterwards) + r, g, b
struct Centroid {
    float v[11];
    explicit Centroid(const kt_point_xyzrgbnormal& p)
    {
        float rgb;
        std::memcpy(&rgb, &p.b, 4);
        const float t[11] = {p.x, p.y, p.z, rgb, p.normal_x, p.normal_y, p.normal_z, p.curvature, (float)p.r, (float)p.g, (float)p.b};
        std::memcpy(v, t, sizeof(v));
    }
    void operator+=(const Centroid& o) { for (int a = 0; a < 11; ++a) v[a] += o.v[a]; }
};

struct LeafRef { unsigned int key, src; };

}  // namespace

extern "C" int kt_host_voxel_grid_normal(const kt_point_xyzrgbnormal* in, size_t n, float leaf, kt_point_xyzrgbnormal* out, size_t* n_out)
{
    KT_ARG(n_out && (n == 0 || (in && out)) && leaf > 0 && n < (1u << 31));
    *n_out = 0;
    if (n == 0) return KT_OK;
    const float inv_leaf = 1.0f / leaf;
    // getMinMax3D
    float lo[3] = {in[0].x, in[0].y, in[0].z}, hi[3] = {in[0].x, in[0].y, in[0].z};
    for (size_t i = 1; i < n; ++i) {
        const float p[3] = {in[i].x, in[i].y, in[i].z};
        for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], p[a]); hi[a] = std::fmax(hi[a], p[a]); }
    }
    // "Leaf size is too small for the input dataset. Integer indices would overflow.": the cloud passes through as it is
    long long cells = 1;
    for (int a = 0; a < 3; ++a) cells *= (long long)((hi[a] - lo[a]) * inv_leaf) + 1;
    if (cells > 2147483647LL) {
        std::memcpy(out, in, n * sizeof(kt_point_xyzrgbnormal));
        *n_out = n;
        return KT_OK;
    }
    int min_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)std::floor(lo[a] * inv_leaf);
        div_b[a] = (int)std::floor(hi[a] * inv_leaf) - min_b[a] + 1;
    }
    std::vector<LeafRef> refs(n);
    for (size_t i = 0; i < n; ++i) {
        const int i0 = (int)(std::floor(in[i].x * inv_leaf) - (float)min_b[0]);
        const int i1 = (int)(std::floor(in[i].y * inv_leaf) - (float)min_b[1]);
        const int i2 = (int)(std::floor(in[i].z * inv_leaf) - (float)min_b[2]);
        refs[i].key = (unsigned int)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]);
        refs[i].src = (unsigned int)i;
    }
    // PCL sorts with std::sort (unstable): the order of a leaf's points is fixed to the input order, as in kt_slice_process
    std::stable_sort(refs.begin(), refs.end(), [](const LeafRef& a, const LeafRef& b) { return a.key < b.key; });
    size_t leaves = 0;
    for (size_t i = 0; i < n;) {
        Centroid c(in[refs[i].src]);
        size_t j = i + 1;
        for (; j < n && refs[j].key == refs[i].key; ++j) c += Centroid(in[refs[j].src]);
        const float inv_count = 1.0f / (float)(j - i);   // Eigen 3.2: `centroid /= count` multiplies by Scalar(1) / count
        for (int a = 0; a < 11; ++a) c.v[a] *= inv_count;
        kt_point_xyzrgbnormal o;
        std::memset(&o, 0, sizeof(o));   // output.points.resize(): value-initialised points (data[3] = 1 below, everything else 0)
        o.x = c.v[0]; o.y = c.v[1]; o.z = c.v[2]; o.pad0 = 1.0f;
        o.normal_x = c.v[4]; o.normal_y = c.v[5]; o.normal_z = c.v[6];
        o.curvature = c.v[7];
        const int rgb = ((int)c.v[8] << 16) | ((int)c.v[9] << 8) | (int)c.v[10];   // "pack r/g/b into rgb": the alpha byte ends up 0
        std::memcpy(&o.b, &rgb, 4);
        out[leaves++] = o;
        i = j;
    }
    *n_out = leaves;
    return KT_OK;
}

// PCDWriter::writeBinary<pcl::PointXYZRGBNormal>: generateHeader's text, "DATA binary\n", then the registered fields of every point
// back to back (32 bytes; the struct's padding floats are not fields).  width = n, height = 1 (an unorganised cloud after insert()),
// sensor origin 0 and identity orientation.
extern "C" int kt_host_save_pcd(const char* path, const kt_point_xyzrgbnormal* pts, size_t n)
{
    KT_ARG(path && (n == 0 || pts));
    FILE* f = std::fopen(path, "wb");
    if (!f) { kt_set_error("cannot open %s for writing", path); return KT_ERR_ARG; }
    const std::string count = std::to_string(n);
    const std::string header = "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb normal_x normal_y normal_z curvature\n"
                               "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH " + count + "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                               "POINTS " + count + "\nDATA binary\n";
    bool ok = std::fwrite(header.data(), 1, header.size(), f) == header.size();
    std::vector<unsigned char> rows;
    const size_t chunk = 1 << 16;
    rows.resize(std::min(n, chunk) * 32);
    for (size_t i0 = 0; ok && i0 < n; i0 += chunk) {
        const size_t m = std::min(chunk, n - i0);
        for (size_t i = 0; i < m; ++i) {
            const kt_point_xyzrgbnormal& p = pts[i0 + i];
            unsigned char* o = &rows[i * 32];
            std::memcpy(o, &p.x, 12);             // x y z
            std::memcpy(o + 12, &p.b, 4);         // rgb (b g r a)
            std::memcpy(o + 16, &p.normal_x, 12); // normal_x normal_y normal_z
            std::memcpy(o + 28, &p.curvature, 4);
        }
        ok = std::fwrite(rows.data(), 32, m, f) == m;
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { kt_set_error("short write to %s", path); return KT_ERR_ARG; }
    return KT_OK;
}
