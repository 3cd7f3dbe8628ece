/*
 * kt_ref_api.cpp -- C entry points of oracle/_ref/libkt_ref.so: each ktref_* function has the argument list of the
 * kto_* function of the same name (oracle/kt_oracle.h), moves the caller's dense host arrays into the reference's own
 * DeviceArray / DeviceArray2D containers (containers/device_memory.cpp, compiled from /root/reference as it is), calls
 * the REFERENCE'S OWN host wrapper from frontend/cuda/internal.h:295-536 -- whose launch geometry, argument packing and
 * kernels are therefore the reference's, executed by the fiber emulation of kt_cuda_emul.h -- and copies the results
 * back.  TEST INFRASTRUCTURE ONLY: loaded by tests/ (oracle-vs-reference pinning, golden generation), nothing else.
 */
#include "internal.h"        /* the reference's, found through -I /root/reference/src/frontend/cuda */

#include <stdint.h>
#include <string.h>

int ktref_volume_side = 512;  /* kt_ref_vol.h: VOLUME_X/Y/Z of the three volume .cu files */

namespace {

template <class T> struct Img : DeviceArray2D<T> {
    Img(int rows, int cols) : DeviceArray2D<T>(rows, cols) {}
    Img(int rows, int cols, const void* host) : DeviceArray2D<T>(rows, cols) { this->upload(host, (size_t)cols * sizeof(T), rows, cols); }
    void get(void* host) const { this->download(host, (size_t)this->cols() * sizeof(T)); }
    /* the DataTerm image is addressed as ONE linear array by both kernels that touch it (reduce.cu:444, :767: corresImg.data[k], k the
     * pixel index) whatever the row pitch of its allocation: its contents are the first rows * cols entries behind the base pointer */
    void put_linear(const void* host) { cudaSafeCall(cudaMemcpy(this->ptr(), host, (size_t)this->rows() * this->cols() * sizeof(T), cudaMemcpyHostToDevice)); }
    void get_linear(void* host) const { cudaSafeCall(cudaMemcpy(host, this->ptr(), (size_t)this->rows() * this->cols() * sizeof(T), cudaMemcpyDeviceToHost)); }
};

Mat33 mat33(const float* m)
{
    Mat33 r;
    for (int i = 0; i < 3; ++i) r.data[i] = make_float3(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
    return r;
}
float3 f3(const float* v) { return make_float3(v[0], v[1], v[2]); }
int3 i3(const int* v) { return make_int3(v[0], v[1], v[2]); }

/* volumes are dense (pitch = N * sizeof(T)): tsdf_volume.cu:612 indexes ptr(0)[flat] */
template <class T> struct Vol {
    T* d;
    size_t bytes;
    void* host;
    Vol(void* h, int N) : d(0), bytes((size_t)N * N * N * sizeof(T)), host(h)
    {
        cudaSafeCall(cudaMalloc((void**)&d, bytes));
        cudaSafeCall(cudaMemcpy(d, h, bytes, cudaMemcpyHostToDevice));
    }
    ~Vol() { cudaFree(d); }
    void back() { cudaSafeCall(cudaMemcpy(host, d, bytes, cudaMemcpyDeviceToHost)); }
    PtrStep<T> view(int N) { return PtrStep<T>(d, (size_t)N * sizeof(T)); }
};

}  // namespace

extern "C" {

typedef struct { float fx, fy, cx, cy; } ktref_intr;
typedef struct { float m[9]; } ktref_mat33;

void ktref_set_volume_side(int N) { ktref_volume_side = N; }
int ktref_sizeof_dataterm(void) { return (int)sizeof(DataTerm); }
int ktref_sizeof_point(void) { return (int)sizeof(PointXYZRGB); }
int ktref_sizeof_jtj(void) { return (int)sizeof(JtJJtrSE3); }
int ktref_max_threads(void) { return MAX_THREADS; }

/* ---- bilateral_pyrdown.cu ---- */
void ktref_bilateral_filter(const uint16_t* src, uint16_t* dst, int cols, int rows)
{
    Img<unsigned short> s(rows, cols, src), d(rows, cols, dst);
    bilateralFilter(s, d);
    d.get(dst);
}

void ktref_pyr_down(const uint16_t* src, int scols, int srows, uint16_t* dst)
{
    Img<unsigned short> s(srows, scols, src), d(srows / 2, scols / 2, dst);
    pyrDown(s, d);
    d.get(dst);
}

void ktref_depth_to_metres(const uint16_t* src, float* dst, int cols, int rows, int cutoff)
{
    Img<unsigned short> s(rows, cols, src);
    Img<float> d(rows, cols, dst);
    shortDepthToMetres(s, d, cutoff);
    d.get(dst);
}

void ktref_bgr_to_intensity(const uint8_t* src_rgb24, uint8_t* dst, int cols, int rows)
{
    Img<PixelRGB> s(rows, cols, src_rgb24);
    Img<unsigned char> d(rows, cols, dst);
    imageBGRToIntensity(s, d);
    d.get(dst);
}

void ktref_pyr_down_gauss_f32(const float* src, int scols, int srows, float* dst)
{
    Img<float> s(srows, scols, src), d(srows / 2, scols / 2, dst);
    pyrDownGaussF(s, d);
    d.get(dst);
}

void ktref_pyr_down_gauss_u8(const uint8_t* src, int scols, int srows, uint8_t* dst)
{
    Img<unsigned char> s(srows, scols, src), d(srows / 2, scols / 2, dst);
    pyrDownUcharGauss(s, d);
    d.get(dst);
}

void ktref_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy)
{
    Img<unsigned char> s(rows, cols, src);
    Img<short> gx(rows, cols, dx), gy(rows, cols, dy);
    computeDerivativeImages(s, gx, gy);
    gx.get(dx);
    gy.get(dy);
}

void ktref_project_to_cloud(const float* depth, int cols, int rows, float* cloud_xyz, double fx, double fy, double cx, double cy, int level)
{
    Img<float> d(rows, cols, depth);
    Img<float3> c(rows, cols, cloud_xyz);
    IntrDoublePrecision K(fx, fy, cx, cy);
    projectToPointCloud(d, c, K, level);
    c.get(cloud_xyz);
}

/* ---- maps.cu ---- */
void ktref_create_vmap(ktref_intr intr, const uint16_t* depth, int cols, int rows, float* vmap)
{
    Img<unsigned short> d(rows, cols, depth);
    Img<float> v(3 * rows, cols, vmap);
    createVMap(Intr(intr.fx, intr.fy, intr.cx, intr.cy), d, v);
    v.get(vmap);
}

void ktref_create_nmap(const float* vmap, int cols, int rows, float* nmap)
{
    Img<float> v(3 * rows, cols, vmap), n(3 * rows, cols, nmap);
    createNMap(v, n);
    n.get(nmap);
}

void ktref_transform_maps(const float* vmap_src, const float* nmap_src, int cols, int rows, const ktref_mat33* R, const float t[3],
                          float* vmap_dst, float* nmap_dst)
{
    Img<float> vs(3 * rows, cols, vmap_src), ns(3 * rows, cols, nmap_src), vd(3 * rows, cols, vmap_dst), nd(3 * rows, cols, nmap_dst);
    tranformMaps(vs, ns, mat33(R->m), f3(t), vd, nd);
    vd.get(vmap_dst);
    nd.get(nmap_dst);
}

void ktref_resize_vmap(const float* in, int in_cols, int in_rows, float* out)
{
    Img<float> i(3 * in_rows, in_cols, in), o(3 * (in_rows / 2), in_cols / 2, out);
    resizeVMap(i, o);
    o.get(out);
}

void ktref_resize_nmap(const float* in, int in_cols, int in_rows, float* out)
{
    Img<float> i(3 * in_rows, in_cols, in), o(3 * (in_rows / 2), in_cols / 2, out);
    resizeNMap(i, o);
    o.get(out);
}

/* ---- image_generator.cu ---- */
void ktref_generate_image(const float* vmap, const float* nmap, const uint8_t* vmap_curr_color, int cols, int rows, const float light_pos[3],
                          int light_number, uint8_t* dst, uint8_t* dst_color)
{
    Img<float> v(3 * rows, cols, vmap), n(3 * rows, cols, nmap);
    Img<uchar4> c(rows, cols, vmap_curr_color);
    Img<uchar3> d(rows, cols, dst), dc(rows, cols, dst_color);
    LightSource light;
    light.pos[0] = f3(light_pos);
    light.number = light_number;
    generateImage(v, n, c, light, d, dc);
    d.get(dst);
    dc.get(dst_color);
}

void ktref_generate_depth(const ktref_mat33* R_inv, const float t[3], const float* vmap, const float* nmap, int cols, int rows, uint16_t* dst,
                          float max_depth)
{
    Img<float> v(3 * rows, cols, vmap), n(3 * rows, cols, nmap);
    Img<unsigned short> d(rows, cols, dst);
    generateDepth(mat33(R_inv->m), f3(t), v, n, d, max_depth);
    d.get(dst);
}

/* ---- reduce.cu ---- */
void ktref_icp_step(const ktref_mat33* Rcurr, const float tcurr[3], const float* vmap_curr, const float* nmap_curr, const ktref_mat33* Rprev_inv,
                    const float tprev[3], ktref_intr intr, const float* vmap_g_prev, const float* nmap_g_prev, int cols, int rows,
                    float dist_thres, float angle_thres, int threads, int blocks, float A[36], float b[6], float residual[2])
{
    Img<float> vc(3 * rows, cols, vmap_curr), nc(3 * rows, cols, nmap_curr), vp(3 * rows, cols, vmap_g_prev), np(3 * rows, cols, nmap_g_prev);
    DeviceArray<JtJJtrSE3> sum(MAX_THREADS), out(1);   /* ICPOdometry.cpp:41-42 */
    icpStep(mat33(Rcurr->m), f3(tcurr), vc, nc, mat33(Rprev_inv->m), f3(tprev), Intr(intr.fx, intr.fy, intr.cx, intr.cy), vp, np, dist_thres,
            angle_thres, sum, out, A, b, residual, threads, blocks);
}

void ktref_rgb_residual(float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* last_depth, const float* next_depth,
                        const uint8_t* last_image, const uint8_t* next_image, int cols, int rows, void* corres, float max_depth_delta,
                        const float kt[3], const ktref_mat33* krkinv, int threads, int blocks, int* sigma_sum, int* count)
{
    Img<short> gx(rows, cols, dIdx), gy(rows, cols, dIdy);
    Img<float> ld(rows, cols, last_depth), nd(rows, cols, next_depth);
    Img<unsigned char> li(rows, cols, last_image), ni(rows, cols, next_image);
    Img<DataTerm> c(rows, cols);
    c.put_linear(corres);
    DeviceArray<int2> sum(MAX_THREADS);                /* RGBDOdometry.cpp:45-47 */
    computeRgbResidual(min_scale, gx, gy, ld, nd, li, ni, c, sum, max_depth_delta, f3(kt), mat33(krkinv->m), *sigma_sum, *count, threads, blocks);
    c.get_linear(corres);
}

void ktref_rgb_step(const void* corres, float sigma, const float* cloud_xyz, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                    float sobel_scale, int cols, int rows, int threads, int blocks, float A[36], float b[6])
{
    Img<DataTerm> c(rows, cols);
    c.put_linear(corres);
    Img<float3> cl(rows, cols, cloud_xyz);
    Img<short> gx(rows, cols, dIdx), gy(rows, cols, dIdy);
    DeviceArray<JtJJtrSE3> sum(MAX_THREADS), out(1);
    rgbStep(c, sigma, cl, fx, fy, gx, gy, sobel_scale, sum, out, A, b, threads, blocks);
}

/* ---- tsdf_volume.cu ---- */
void ktref_init_volume(int16_t* vol, int N)
{
    ktref_volume_side = N;
    Vol<short> v(vol, N);
    initVolume(v.view(N));
    v.back();
}

void ktref_init_color_volume(uint8_t* cvol, int N)
{
    ktref_volume_side = N;
    Vol<uchar4> v(cvol, N);
    initColorVolume(v.view(N));
    v.back();
}

void ktref_integrate_tsdf(const uint16_t* depth_raw, int cols, int rows, ktref_intr intr, const float volume_size[3], const ktref_mat33* Rcurr_inv,
                          const float tcurr[3], float tranc_dist, int16_t* volume, float* depth_scaled, const int voxel_wrap[3],
                          uint8_t* color_volume, const uint8_t* colors_rgb24, const float* nmap_curr, int angle_color, int N)
{
    ktref_volume_side = N;
    Img<unsigned short> d(rows, cols, depth_raw);
    Img<uchar3> rgb(rows, cols, colors_rgb24);
    Img<float> n(3 * rows, cols, nmap_curr);
    DeviceArray2D<float> scaled;
    Vol<short> v(volume, N);
    Vol<uchar4> c(color_volume, N);
    integrateTsdfVolume(d, Intr(intr.fx, intr.fy, intr.cx, intr.cy), f3(volume_size), mat33(Rcurr_inv->m), f3(tcurr), tranc_dist, v.view(N),
                        scaled, i3(voxel_wrap), c.view(N), rgb, n, angle_color != 0);
    v.back();
    c.back();
    if (depth_scaled) scaled.download(depth_scaled, (size_t)cols * sizeof(float));
}

void ktref_raycast(ktref_intr intr, const ktref_mat33* Rcurr, const float tcurr[3], float tranc_dist, const float volume_size[3],
                   const int16_t* volume, float* vmap, float* nmap, int cols, int rows, const int voxel_wrap[3], uint8_t* vmap_curr_color,
                   const uint8_t* color_volume, int N)
{
    ktref_volume_side = N;
    Vol<short> v((void*)volume, N);
    Vol<uchar4> c((void*)color_volume, N);
    Img<float> vm(3 * rows, cols, vmap), nm(3 * rows, cols, nmap);
    Img<uchar4> col(rows, cols, vmap_curr_color);
    raycast(Intr(intr.fx, intr.fy, intr.cx, intr.cy), mat33(Rcurr->m), f3(tcurr), tranc_dist, f3(volume_size), v.view(N), vm, nm, i3(voxel_wrap), col,
            c.view(N));
    vm.get(vmap);
    nm.get(nmap);
    col.get(vmap_curr_color);
}

void ktref_clear_volume(void* vol, int elem_size, int N, int axis, int back, int current_wrap, int delta_wrap)
{
    ktref_volume_side = N;
    if (elem_size == 2) {
        Vol<short> v(vol, N);
        PtrStep<short> a = v.view(N);
        if (axis == 0) (back ? clearVolumeXBack : clearVolumeX)(a, current_wrap, delta_wrap);
        else if (axis == 1) (back ? clearVolumeYBack : clearVolumeY)(a, current_wrap, delta_wrap);
        else (back ? clearVolumeZBack : clearVolumeZ)(a, current_wrap, delta_wrap);
        v.back();
    } else {
        Vol<uchar4> v(vol, N);
        PtrStep<uchar4> a = v.view(N);
        if (axis == 0) (back ? clearVolumeXBackc : clearVolumeXc)(a, current_wrap, delta_wrap);
        else if (axis == 1) (back ? clearVolumeYBackc : clearVolumeYc)(a, current_wrap, delta_wrap);
        else (back ? clearVolumeZBackc : clearVolumeZc)(a, current_wrap, delta_wrap);
        v.back();
    }
}

/* ---- extract.cu ---- */
size_t ktref_extract_cloud_slice(const int16_t* volume, const float volume_size[3], void* out, size_t out_cap, const int voxel_wrap[3],
                                 const uint8_t* color_volume, int minX, int maxX, int minY, int maxY, int minZ, int maxZ, int subsample,
                                 const int real_voxel_wrap[3], int N)
{
    ktref_volume_side = N;
    Vol<short> v((void*)volume, N);
    Vol<uchar4> c((void*)color_volume, N);
    DeviceArray<PointXYZRGB> cloud(out_cap);
    cudaSafeCall(cudaMemset(cloud.ptr(), 0, out_cap * sizeof(PointXYZRGB)));
    PtrStep<uchar4> cv = c.view(N);
    size_t n = extractCloudSlice(v.view(N), f3(volume_size), PtrSz<PointXYZRGB>(cloud.ptr(), out_cap), i3(voxel_wrap), cv, minX, maxX, minY, maxY,
                                 minZ, maxZ, subsample, i3(real_voxel_wrap));
    cudaSafeCall(cudaMemcpy(out, cloud.ptr(), out_cap * sizeof(PointXYZRGB), cudaMemcpyDeviceToHost));
    return n;
}

}  // extern "C"
