// kt_context.hip -- context, error reporting and device memory for libkt_hip.so.
// Replaces the reference's containers/device_memory.cpp (ref-counted cudaMalloc / cudaMallocPitch),
// containers/initialization.cpp and the cudaSafeCall error path (internal.h:76-86).
// Memory is dense (no pitch): the reference's volume kernels already assume pitch == cols*sizeof(T).
#include "kt_common.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void kt_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int kt_check(hipError_t e, const char* what, const char* file, int line)
{
    if (e == hipSuccess) return KT_OK;
    kt_set_error("%s\t%s:%d (%s)", hipGetErrorString(e), file, line, what);
    return e == hipErrorOutOfMemory ? KT_ERR_NOMEM : KT_ERR_HIP;
}

extern "C" {

const char* kt_last_error(void) { return g_err; }
const char* kt_version(void) { return "kintinuous_amd 0.1 (gfx950)"; }

int kt_device_count(int* count)
{
    KT_ARG(count);
    KT_HIP(hipGetDeviceCount(count));
    return KT_OK;
}

int kt_ctx_create(int device, kt_ctx** out)
{
    KT_ARG(out);
    KT_HIP(hipSetDevice(device));
    kt_ctx* c = new kt_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    int s = kt_check(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate", __FILE__, __LINE__);
    if (s != KT_OK) { delete c; return s; }
    c->own_stream = true;
    c->red_max_blocks = 2048;   // 512 KB of hand-off granules: the reduction sets, the residual words, the level kernel's two sets (kt_track.hip)
    KT_HIP(hipMalloc((void**)&c->red_partials, sizeof(double) * 32 * c->red_max_blocks));
    KT_HIP(hipMalloc((void**)&c->red_out, sizeof(float) * 64));
    KT_HIP(hipMalloc((void**)&c->counters, sizeof(unsigned int) * 16));
    KT_HIP(hipMemsetAsync(c->red_partials, 0xff, sizeof(double) * 32 * c->red_max_blocks, c->stream));   // the reduction granules' sentinel (kt_track.hip)
    KT_HIP(hipMemsetAsync(c->counters, 0, sizeof(unsigned int) * 16, c->stream));
    KT_HIP(hipMalloc((void**)&c->pose_gran, 256));
    KT_HIP(hipMemsetAsync(c->pose_gran, 0, 256, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    KT_HIP(hipHostMalloc((void**)&c->red_out_host, sizeof(float) * 64, hipHostMallocDefault));
    KT_HIP(hipHostMalloc((void**)&c->int_out_host, sizeof(int) * 16, hipHostMallocDefault));
    *out = c;
    return KT_OK;
}

int kt_ctx_destroy(kt_ctx* c)
{
    if (!c) return KT_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    kt_integrate_scratch_free(c);
    if (c->slice_ws) (void)kt_slice_ws_destroy((kt_slice_ws*)c->slice_ws);
    (void)hipFree(c->bil_lut);
    (void)hipFree(c->track_state);
    (void)hipFree(c->red_partials);
    (void)hipFree(c->pose_gran);
    (void)hipFree(c->red_out);
    (void)hipFree(c->counters);
    (void)hipHostFree(c->red_out_host);
    (void)hipHostFree(c->int_out_host);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return KT_OK;
}

int kt_ctx_set_stream(kt_ctx* c, void* hip_stream)
{
    KT_ARG(c);
    KT_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = (hipStream_t)hip_stream;
    return KT_OK;
}

void* kt_ctx_stream(kt_ctx* c) { return c ? (void*)c->stream : nullptr; }

int kt_sync(kt_ctx* c)
{
    KT_ARG(c);
    KT_HIP(hipStreamSynchronize(c->stream));
    return KT_OK;
}

int kt_malloc(kt_ctx* c, size_t bytes, void** dptr)
{
    KT_ARG(c && dptr);
    KT_HIP(hipSetDevice(c->device));
    KT_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return KT_OK;
}

int kt_free(kt_ctx* c, void* dptr)
{
    KT_ARG(c);
    if (dptr) KT_HIP(hipFree(dptr));
    return KT_OK;
}

int kt_memset(kt_ctx* c, void* dptr, int value, size_t bytes)
{
    KT_ARG(c && dptr);
    KT_HIP(hipMemsetAsync(dptr, value, bytes, c->stream));
    return KT_OK;
}

int kt_upload(kt_ctx* c, void* dst, const void* src_host, size_t bytes)
{
    KT_ARG(c && dst && src_host);
    KT_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));  // blocking, like DeviceMemory::upload
    return KT_OK;
}

int kt_download(kt_ctx* c, void* dst_host, const void* src, size_t bytes)
{
    KT_ARG(c && dst_host && src);
    KT_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    return KT_OK;
}

int kt_upload2d(kt_ctx* c, void* dst, const void* src_host, size_t host_pitch, size_t row_bytes, int rows)
{
    KT_ARG(c && dst && src_host && host_pitch >= row_bytes && rows >= 0);
    KT_HIP(hipMemcpy2DAsync(dst, row_bytes, src_host, host_pitch, row_bytes, (size_t)rows, hipMemcpyHostToDevice, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    return KT_OK;
}

int kt_download2d(kt_ctx* c, void* dst_host, size_t host_pitch, const void* src, size_t row_bytes, int rows)
{
    KT_ARG(c && dst_host && src && host_pitch >= row_bytes && rows >= 0);
    KT_HIP(hipMemcpy2DAsync(dst_host, host_pitch, src, row_bytes, row_bytes, (size_t)rows, hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    return KT_OK;
}

}  // extern "C"
