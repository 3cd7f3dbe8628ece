// internal.h -- the reference's device operator API (frontend/cuda/internal.h:295-536) re-created as inline wrappers
// over libkt_hip.so's C-ABI (include/kt_abi.h).  Same names, argument order and meaning; errors print and exit like
// cudaSafeCall.  Differences: PtrStep<T>/PtrStepSz<T> parameters are taken as DeviceArray2D<T> (the volume is a
// DeviceArray2D<short>(N*N, N), as in TSDFVolume.cpp:52-57), VOL is read from the array's shape, and the reduction
// scratch arrays (sum / out / threads / blocks) are accepted and ignored (they live in the kt_ctx).
#pragma once

#include <stdint.h>

#include "containers/device_array.hpp"

struct float3 { float x, y, z; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
inline int3 make_int3(int x, int y, int z) { int3 r = {x, y, z}; return r; }

typedef kt_intr Intr_pod;
struct Intr {  // internal.h:249-260
    float fx, fy, cx, cy;
    Intr() : fx(0), fy(0), cx(0), cy(0) {}
    Intr(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
    Intr operator()(int level_index) const
    {
        const int div = 1 << level_index;
        return Intr(fx / div, fy / div, cx / div, cy / div);
    }
};
struct IntrDoublePrecision {  // internal.h:262-273
    double fx, fy, cx, cy;
    IntrDoublePrecision() : fx(0), fy(0), cx(0), cy(0) {}
    IntrDoublePrecision(double fx_, double fy_, double cx_, double cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
};
struct Mat33 { float3 data[3]; };  // internal.h:279-282
typedef kt_jtj JtJJtrSE3;
typedef kt_dataterm DataTerm;
typedef kt_point_xyzrgb PointXYZRGB;
struct PixelRGB { unsigned char r, g, b; };  // internal.h:186-189

static_assert(sizeof(Mat33) == sizeof(kt_mat33), "Mat33 layout");
static_assert(sizeof(Intr) == sizeof(kt_intr), "Intr layout");
// internal.h:186-229 == pcl::PointXYZRGBNormal (48 bytes): what CloudSliceProcessor fills processedCloud with
struct PointXYZRGBNormal {
    float x, y, z, pad0;
    float normal_x, normal_y, normal_z, pad1;
    union {
        struct { unsigned char b, g, r, a; };
        float rgb;
        int rgba;
    };
    float curvature;
    float pad2[2];
};
static_assert(sizeof(PointXYZRGB) == 32 && sizeof(PointXYZRGBNormal) == 48 && sizeof(DataTerm) == 16 && sizeof(JtJJtrSE3) == 116, "boundary struct layout");

namespace kt {
inline const kt_intr* abi(const Intr& i) { return reinterpret_cast<const kt_intr*>(&i); }
inline const kt_mat33* abi(const Mat33& m) { return reinterpret_cast<const kt_mat33*>(&m); }
inline const float* abi(const float3& v) { return &v.x; }
inline const int* abi(const int3& v) { return &v.x; }
// VOL of a DeviceArray2D<T>(N*N, N) volume
template <class T> inline int volN(const DeviceArray2D<T>& v) { return v.cols(); }
}  // namespace kt

#define KT_CTX ::kt::device::context()

// ---- image side -------------------------------------------------------------------------------------------------
inline void bilateralFilter(const DeviceArray2D<unsigned short>& src, DeviceArray2D<unsigned short>& dst)
{
    dst.create(src.rows(), src.cols());
    ktSafeCall(kt_bilateral_filter(KT_CTX, src.ptr(), dst.ptr(), src.cols(), src.rows()));
}
inline void pyrDown(const DeviceArray2D<unsigned short>& src, DeviceArray2D<unsigned short>& dst)
{
    dst.create(src.rows() / 2, src.cols() / 2);
    ktSafeCall(kt_pyr_down(KT_CTX, src.ptr(), src.cols(), src.rows(), dst.ptr()));
}
inline void pyrDownGaussF(const DeviceArray2D<float>& src, DeviceArray2D<float>& dst)
{
    dst.create(src.rows() / 2, src.cols() / 2);
    ktSafeCall(kt_pyr_down_gauss_f32(KT_CTX, src.ptr(), src.cols(), src.rows(), dst.ptr()));
}
inline void pyrDownUcharGauss(const DeviceArray2D<unsigned char>& src, DeviceArray2D<unsigned char>& dst)
{
    dst.create(src.rows() / 2, src.cols() / 2);
    ktSafeCall(kt_pyr_down_gauss_u8(KT_CTX, src.ptr(), src.cols(), src.rows(), dst.ptr()));
}
inline void computeDerivativeImages(DeviceArray2D<unsigned char>& src, DeviceArray2D<short>& dx, DeviceArray2D<short>& dy)
{
    dx.create(src.rows(), src.cols());
    dy.create(src.rows(), src.cols());
    ktSafeCall(kt_derivative_images(KT_CTX, src.ptr(), src.cols(), src.rows(), dx.ptr(), dy.ptr()));
}
inline void shortDepthToMetres(const DeviceArray2D<unsigned short>& src, DeviceArray2D<float>& dst, int cutOff)
{
    dst.create(src.rows(), src.cols());
    ktSafeCall(kt_depth_to_metres(KT_CTX, src.ptr(), dst.ptr(), src.cols(), src.rows(), cutOff));
}
inline void imageBGRToIntensity(const DeviceArray2D<PixelRGB>& src, DeviceArray2D<unsigned char>& dst)
{
    dst.create(src.rows(), src.cols());
    ktSafeCall(kt_bgr_to_intensity(KT_CTX, reinterpret_cast<const uint8_t*>(src.ptr()), dst.ptr(), src.cols(), src.rows()));
}
inline void projectToPointCloud(const DeviceArray2D<float>& depth, const DeviceArray2D<float3>& cloud,
                                IntrDoublePrecision& intrinsics, const int& level)
{
    ktSafeCall(kt_project_to_cloud(KT_CTX, depth.ptr(), depth.cols(), depth.rows(),
                                   const_cast<float*>(&cloud.ptr()->x), intrinsics.fx, intrinsics.fy, intrinsics.cx, intrinsics.cy, level));
}
inline void createVMap(const Intr& intr, const DeviceArray2D<unsigned short>& depth, DeviceArray2D<float>& vmap)
{
    vmap.create(depth.rows() * 3, depth.cols());
    ktSafeCall(kt_create_vmap(KT_CTX, kt::abi(intr), depth.ptr(), depth.cols(), depth.rows(), vmap.ptr()));
}
inline void createNMap(const DeviceArray2D<float>& vmap, DeviceArray2D<float>& nmap)
{
    nmap.create(vmap.rows(), vmap.cols());
    ktSafeCall(kt_create_nmap(KT_CTX, vmap.ptr(), vmap.cols(), vmap.rows() / 3, nmap.ptr()));
}
inline void tranformMaps(const DeviceArray2D<float>& vmap_src, const DeviceArray2D<float>& nmap_src, const Mat33& Rmat,
                         const float3& tvec, DeviceArray2D<float>& vmap_dst, DeviceArray2D<float>& nmap_dst)
{
    vmap_dst.create(vmap_src.rows(), vmap_src.cols());
    nmap_dst.create(vmap_src.rows(), vmap_src.cols());
    ktSafeCall(kt_transform_maps(KT_CTX, vmap_src.ptr(), nmap_src.ptr(), vmap_src.cols(), vmap_src.rows() / 3, kt::abi(Rmat),
                                 kt::abi(tvec), vmap_dst.ptr(), nmap_dst.ptr()));
}
inline void resizeVMap(const DeviceArray2D<float>& input, DeviceArray2D<float>& output)
{
    output.create(input.rows() / 2, input.cols() / 2);  // rows = 3 * (in_rows / 2)
    ktSafeCall(kt_resize_vmap(KT_CTX, input.ptr(), input.cols(), input.rows() / 3, output.ptr()));
}
inline void resizeNMap(const DeviceArray2D<float>& input, DeviceArray2D<float>& output)
{
    output.create(input.rows() / 2, input.cols() / 2);
    ktSafeCall(kt_resize_nmap(KT_CTX, input.ptr(), input.cols(), input.rows() / 3, output.ptr()));
}

// ---- volume -----------------------------------------------------------------------------------------------------
inline void initVolume(DeviceArray2D<short>& volume) { ktSafeCall(kt_init_volume(KT_CTX, volume.ptr(), kt::volN(volume))); }
inline void initColorVolume(DeviceArray2D<uchar4>& color_volume)
{
    ktSafeCall(kt_init_color_volume(KT_CTX, &color_volume.ptr()->x, kt::volN(color_volume)));
}
#define KT_CLEAR_WRAPPER(NAME, AXIS, BACK)                                                                            \
    inline void NAME(DeviceArray2D<short>& array, const int currentVoxelWrap, const int deltaVoxelWrap)               \
    {                                                                                                                 \
        ktSafeCall(kt_clear_volume(KT_CTX, array.ptr(), 2, kt::volN(array), AXIS, BACK, currentVoxelWrap, deltaVoxelWrap)); \
    }                                                                                                                 \
    inline void NAME##c(DeviceArray2D<uchar4>& array, const int currentVoxelWrap, const int deltaVoxelWrap)           \
    {                                                                                                                 \
        ktSafeCall(kt_clear_volume(KT_CTX, array.ptr(), 4, kt::volN(array), AXIS, BACK, currentVoxelWrap, deltaVoxelWrap)); \
    }
KT_CLEAR_WRAPPER(clearVolumeX, 0, 0)
KT_CLEAR_WRAPPER(clearVolumeXBack, 0, 1)
KT_CLEAR_WRAPPER(clearVolumeY, 1, 0)
KT_CLEAR_WRAPPER(clearVolumeYBack, 1, 1)
KT_CLEAR_WRAPPER(clearVolumeZ, 2, 0)
KT_CLEAR_WRAPPER(clearVolumeZBack, 2, 1)
#undef KT_CLEAR_WRAPPER

inline void integrateTsdfVolume(const DeviceArray2D<unsigned short>& depth_raw, const Intr& intr, const float3& volume_size,
                                const Mat33& Rcurr_inv, const float3& tcurr, float tranc_dist, DeviceArray2D<short>& volume,
                                DeviceArray2D<float>& depthRawScaled, const int3& voxelWrap, DeviceArray2D<uchar4>& color_volume,
                                const DeviceArray2D<PixelRGB>& colors, const DeviceArray2D<float>& nmap_curr, bool angleColor)
{
    depthRawScaled.create(depth_raw.rows(), depth_raw.cols());
    ktSafeCall(kt_integrate_tsdf(KT_CTX, depth_raw.ptr(), depth_raw.cols(), depth_raw.rows(), kt::abi(intr), kt::abi(volume_size),
                                 kt::abi(Rcurr_inv), kt::abi(tcurr), tranc_dist, volume.ptr(), depthRawScaled.ptr(),
                                 kt::abi(voxelWrap), &color_volume.ptr()->x, reinterpret_cast<const uint8_t*>(colors.ptr()),
                                 nmap_curr.ptr(), angleColor ? 1 : 0, kt::volN(volume)));
}
inline void raycast(const Intr& intr, const Mat33& Rcurr, const float3& tcurr, float tranc_dist, const float3& volume_size,
                    const DeviceArray2D<short>& volume, DeviceArray2D<float>& vmap, DeviceArray2D<float>& nmap,
                    const int3& voxelWrap, DeviceArray2D<uchar4>& vmap_curr_color, DeviceArray2D<uchar4>& color_volume)
{
    const int cols = vmap.cols(), rows = vmap.rows() / 3;
    ktSafeCall(kt_raycast(KT_CTX, kt::abi(intr), kt::abi(Rcurr), kt::abi(tcurr), tranc_dist, kt::abi(volume_size), volume.ptr(),
                          vmap.ptr(), nmap.ptr(), cols, rows, kt::abi(voxelWrap), &vmap_curr_color.ptr()->x,
                          &color_volume.ptr()->x, kt::volN(volume)));
}
// internal.h:284-293, 435-442: view products (KintinuousTracker::getImage / getModelDepth)
struct LightSource { float3 pos[1]; int number; };
inline void generateImage(const DeviceArray2D<float>& vmap, const DeviceArray2D<float>& nmap, const DeviceArray2D<uchar4>& vmap_curr_color,
                          const LightSource& light, DeviceArray2D<PixelRGB>& dst, DeviceArray2D<PixelRGB>& dstColor)
{
    const int cols = vmap.cols(), rows = vmap.rows() / 3;
    dst.create(rows, cols);
    dstColor.create(rows, cols);
    ktSafeCall(kt_generate_image(KT_CTX, vmap.ptr(), nmap.ptr(), &vmap_curr_color.ptr()->x, cols, rows, kt::abi(light.pos[0]), light.number,
                                 &dst.ptr()->r, &dstColor.ptr()->r));
}
inline void generateDepth(const Mat33& R_inv, const float3& t, const DeviceArray2D<float>& vmap, const DeviceArray2D<float>& nmap,
                          DeviceArray2D<unsigned short>& dst, float /*maxDepth*/)
{
    const int cols = vmap.cols(), rows = vmap.rows() / 3;
    dst.create(rows, cols);
    ktSafeCall(kt_generate_depth(KT_CTX, kt::abi(R_inv), kt::abi(t), vmap.ptr(), nmap.ptr(), cols, rows, dst.ptr()));
}
inline int GetGridDim(int D, int B) { return (D + B - 1) / B; }  // internal.h:458 (launch arithmetic; nothing here needs it)

inline size_t extractCloudSlice(const DeviceArray2D<short>& volume, const float3& volume_size, DeviceArray<PointXYZRGB>& output,
                                int3 voxelWrap, DeviceArray2D<uchar4>& color_volume, int minX, int maxX, int minY, int maxY,
                                int minZ, int maxZ, int subsample, int3 realVoxelWrap)
{
    size_t count = 0;
    ktSafeCall(kt_extract_cloud_slice(KT_CTX, volume.ptr(), kt::abi(volume_size), output.ptr(), output.size(), kt::abi(voxelWrap),
                                      &color_volume.ptr()->x, minX, maxX, minY, maxY, minZ, maxZ, subsample,
                                      kt::abi(realVoxelWrap), kt::volN(volume), &count));
    return count;
}

// ---- the same operators with the reference's own parameter lists (internal.h:349-470): PtrStep / PtrStepSz / PtrSz views, so that
// call sites written against the reference -- tsdf_volume_->data() and color_volume_->data() returned BY VALUE, the colour volume a
// DeviceArray2D<int> (ColorVolume.cpp:80, TSDFVolume.cpp:100, KintinuousTracker.cpp:504-521, 677-809, 864-890) -- compile as they are.
// A view carries no size: the volume side is Volume::get().getResolution(), the reference's compile-time VOL.  Rows must be dense.
namespace kt {
inline int volSide();   // Volume.h (included after this header by every user of the volume operators)
template <class T> inline T* dense(const PtrStep<T>& v, size_t row_bytes)
{
    if (v.step != row_bytes) { std::fprintf(stderr, "Error: view with padded rows (step %zu, row %zu)\t%s:%d\n", v.step, row_bytes, __FILE__, __LINE__); std::exit(0); }
    return const_cast<T*>(v.data);
}
}  // namespace kt
inline void initVolume(PtrStep<short> array) { const int N = kt::volSide(); ktSafeCall(kt_init_volume(KT_CTX, kt::dense(array, (size_t)N * 2), N)); }
inline void initColorVolume(PtrStep<uchar4> array)
{
    const int N = kt::volSide();
    ktSafeCall(kt_init_color_volume(KT_CTX, &kt::dense(array, (size_t)N * 4)->x, N));
}
#define KT_CLEAR_VIEW(NAME, AXIS, BACK)                                                                                              \
    inline void NAME(PtrStep<short> array, const int currentVoxelWrap, const int deltaVoxelWrap)                                     \
    {                                                                                                                                \
        const int N = kt::volSide();                                                                                                 \
        ktSafeCall(kt_clear_volume(KT_CTX, kt::dense(array, (size_t)N * 2), 2, N, AXIS, BACK, currentVoxelWrap, deltaVoxelWrap));    \
    }                                                                                                                                \
    inline void NAME##c(PtrStep<uchar4> array, const int currentVoxelWrap, const int deltaVoxelWrap)                                 \
    {                                                                                                                                \
        const int N = kt::volSide();                                                                                                 \
        ktSafeCall(kt_clear_volume(KT_CTX, kt::dense(array, (size_t)N * 4), 4, N, AXIS, BACK, currentVoxelWrap, deltaVoxelWrap));    \
    }
KT_CLEAR_VIEW(clearVolumeX, 0, 0)
KT_CLEAR_VIEW(clearVolumeXBack, 0, 1)
KT_CLEAR_VIEW(clearVolumeY, 1, 0)
KT_CLEAR_VIEW(clearVolumeYBack, 1, 1)
KT_CLEAR_VIEW(clearVolumeZ, 2, 0)
KT_CLEAR_VIEW(clearVolumeZBack, 2, 1)
#undef KT_CLEAR_VIEW
inline void integrateTsdfVolume(const PtrStepSz<unsigned short>& depth_raw, const Intr& intr, const float3& volume_size, const Mat33& Rcurr_inv,
                                const float3& tcurr, float tranc_dist, PtrStep<short> volume, DeviceArray2D<float>& depthRawScaled,
                                const int3& voxelWrap, PtrStep<uchar4> color_volume, PtrStepSz<uchar3> colors,
                                const DeviceArray2D<float>& nmap_curr, bool angleColor)
{
    const int N = kt::volSide();
    depthRawScaled.create(depth_raw.rows, depth_raw.cols);
    ktSafeCall(kt_integrate_tsdf(KT_CTX, kt::dense(depth_raw, (size_t)depth_raw.cols * 2), depth_raw.cols, depth_raw.rows, kt::abi(intr),
                                 kt::abi(volume_size), kt::abi(Rcurr_inv), kt::abi(tcurr), tranc_dist, kt::dense(volume, (size_t)N * 2),
                                 depthRawScaled.ptr(), kt::abi(voxelWrap), &kt::dense(color_volume, (size_t)N * 4)->x,
                                 &kt::dense(colors, (size_t)depth_raw.cols * 3)->x, nmap_curr.ptr(), angleColor ? 1 : 0, N));
}
inline void raycast(const Intr& intr, const Mat33& Rcurr, const float3& tcurr, float tranc_dist, const float3& volume_size,
                    const PtrStep<short>& volume, DeviceArray2D<float>& vmap, DeviceArray2D<float>& nmap, const int3& voxelWrap,
                    DeviceArray2D<uchar4>& vmap_curr_color, PtrStep<uchar4> color_volume)
{
    const int N = kt::volSide(), cols = vmap.cols(), rows = vmap.rows() / 3;
    ktSafeCall(kt_raycast(KT_CTX, kt::abi(intr), kt::abi(Rcurr), kt::abi(tcurr), tranc_dist, kt::abi(volume_size), kt::dense(volume, (size_t)N * 2),
                          vmap.ptr(), nmap.ptr(), cols, rows, kt::abi(voxelWrap), &vmap_curr_color.ptr()->x,
                          &kt::dense(color_volume, (size_t)N * 4)->x, N));
}
inline void generateImage(const DeviceArray2D<float>& vmap, const DeviceArray2D<float>& nmap, const DeviceArray2D<uchar4>& vmap_curr_color,
                          const LightSource& light, PtrStepSz<uchar3> dst, PtrStepSz<uchar3> dstColor)
{
    const int cols = vmap.cols(), rows = vmap.rows() / 3;
    ktSafeCall(kt_generate_image(KT_CTX, vmap.ptr(), nmap.ptr(), &vmap_curr_color.ptr()->x, cols, rows, kt::abi(light.pos[0]), light.number,
                                 &kt::dense(dst, (size_t)cols * 3)->x, &kt::dense(dstColor, (size_t)cols * 3)->x));
}
inline size_t extractCloudSlice(const PtrStep<short>& volume, const float3& volume_size, PtrSz<PointXYZRGB> output, int3 voxelWrap,
                                PtrStep<uchar4>& color_volume, int minX, int maxX, int minY, int maxY, int minZ, int maxZ, int subsample,
                                int3 realVoxelWrap)
{
    const int N = kt::volSide();
    size_t count = 0;
    ktSafeCall(kt_extract_cloud_slice(KT_CTX, kt::dense(volume, (size_t)N * 2), kt::abi(volume_size), output.data, output.size, kt::abi(voxelWrap),
                                      &kt::dense(color_volume, (size_t)N * 4)->x, minX, maxX, minY, maxY, minZ, maxZ, subsample,
                                      kt::abi(realVoxelWrap), N, &count));
    return count;
}

// ---- tracking reductions ----------------------------------------------------------------------------------------
inline void icpStep(const Mat33& Rcurr, const float3& tcurr, const DeviceArray2D<float>& vmap_curr,
                    const DeviceArray2D<float>& nmap_curr, const Mat33& Rprev_inv, const float3& tprev, const Intr& intr,
                    const DeviceArray2D<float>& vmap_g_prev, const DeviceArray2D<float>& nmap_g_prev, float distThres,
                    float angleThres, DeviceArray<JtJJtrSE3>& /*sum*/, DeviceArray<JtJJtrSE3>& /*out*/, float* matrixA_host,
                    float* vectorB_host, float* residual_host, int /*threads*/, int /*blocks*/)
{
    ktSafeCall(kt_icp_step(KT_CTX, kt::abi(Rcurr), kt::abi(tcurr), vmap_curr.ptr(), nmap_curr.ptr(), kt::abi(Rprev_inv),
                           kt::abi(tprev), kt::abi(intr), vmap_g_prev.ptr(), nmap_g_prev.ptr(), vmap_curr.cols(),
                           vmap_curr.rows() / 3, distThres, angleThres, matrixA_host, vectorB_host, residual_host));
}
inline void rgbStep(const DeviceArray2D<DataTerm>& corresImg, const float& sigma, const DeviceArray2D<float3>& cloud,
                    const float& fx, const float& fy, const DeviceArray2D<short>& dIdx, const DeviceArray2D<short>& dIdy,
                    const float& sobelScale, DeviceArray<JtJJtrSE3>& /*sum*/, DeviceArray<JtJJtrSE3>& /*out*/,
                    float* matrixA_host, float* vectorB_host, int /*threads*/, int /*blocks*/)
{
    ktSafeCall(kt_rgb_step(KT_CTX, corresImg.ptr(), sigma, &cloud.ptr()->x, fx, fy, dIdx.ptr(), dIdy.ptr(), sobelScale,
                           corresImg.cols(), corresImg.rows(), matrixA_host, vectorB_host));
}
inline void computeRgbResidual(const float& minScale, const DeviceArray2D<short>& dIdx, const DeviceArray2D<short>& dIdy,
                               const DeviceArray2D<float>& lastDepth, const DeviceArray2D<float>& nextDepth,
                               const DeviceArray2D<unsigned char>& lastImage, const DeviceArray2D<unsigned char>& nextImage,
                               DeviceArray2D<DataTerm>& corresImg, DeviceArray<int2>& /*sumResidual*/, const float maxDepthDelta,
                               const float3& kt_, const Mat33& krkinv, int& sigmaSum, int& count, int /*threads*/, int /*blocks*/)
{
    ktSafeCall(kt_rgb_residual(KT_CTX, minScale, dIdx.ptr(), dIdy.ptr(), lastDepth.ptr(), nextDepth.ptr(), lastImage.ptr(),
                               nextImage.ptr(), nextImage.cols(), nextImage.rows(), corresImg.ptr(), maxDepthDelta,
                               kt::abi(kt_), kt::abi(krkinv), &sigmaSum, &count));
}
#undef KT_CTX
