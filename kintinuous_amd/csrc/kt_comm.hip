// kt_comm.hip -- the one collective of the multi-stream layout: every rank (one tracker per GPU, one process per
// rank) contributes its last k dense poses -- k x 16 floats, row-major [R | currentGlobalCamera], KintinuousTracker.h:151-169 -- to
// ONE ncclAllGather over RCCL / xGMI, on a dedicated stream so it never queues behind a frame.  64 B - 2 KB messages: latency
// bound, no ring all-reduce, no data-path collective anywhere else.
// librccl is opened lazily (dlopen) so that libkt_hip.so loads on hosts without it; the calls go through rccl.h's own prototypes.
#include "kt_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

struct kt_comm {
    kt_ctx* ctx;
    ncclComm_t comm;
    hipStream_t stream;
    hipEvent_t poses_ready;
    int rank, nranks;
    float* send; float* recv;   // device staging, grown on demand
    size_t cap_floats;
};

namespace {
struct Rccl {
    void* lib;
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclAllGather) AllGather;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclGetErrorString) GetErrorString;
} g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

int rccl_load()
{
    if (g_rccl.lib) return KT_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { kt_set_error("kt_comm: cannot open librccl.so (%s)", dlerror()); return KT_ERR_STATE; }
#define KT_SYM(field, name) do { g_rccl.field = (decltype(g_rccl.field))dlsym(h, name); if (!g_rccl.field) { kt_set_error("kt_comm: %s missing in librccl", name); dlclose(h); return KT_ERR_STATE; } } while (0)
    KT_SYM(GetUniqueId, "ncclGetUniqueId");
    KT_SYM(CommInitRank, "ncclCommInitRank");
    KT_SYM(AllGather, "ncclAllGather");
    KT_SYM(CommDestroy, "ncclCommDestroy");
    KT_SYM(GetErrorString, "ncclGetErrorString");
#undef KT_SYM
    g_rccl.lib = h;
    return KT_OK;
}
}  // namespace

#define KT_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { kt_set_error("%s\t%s:%d (%s)", g_rccl.GetErrorString(r_), __FILE__, __LINE__, #expr); return KT_ERR_HIP; } } while (0)

extern "C" {

int kt_comm_unique_id(unsigned char id[KT_COMM_ID_BYTES])
{
    KT_ARG(id);
    static_assert(KT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "kt_abi.h and rccl.h disagree on the id size");
    KT_TRY(rccl_load());
    ncclUniqueId u;
    KT_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, KT_COMM_ID_BYTES);
    return KT_OK;
}

int kt_comm_init(kt_ctx* ctx, int rank, int nranks, const unsigned char id[KT_COMM_ID_BYTES], kt_comm** out)
{
    KT_ARG(ctx && id && out && nranks >= 1 && rank >= 0 && rank < nranks);
    KT_TRY(rccl_load());
    KT_HIP(hipSetDevice(ctx->device));
    kt_comm* c = new kt_comm();
    memset(c, 0, sizeof(*c));
    c->ctx = ctx; c->rank = rank; c->nranks = nranks;
    ncclUniqueId u;
    memcpy(u.internal, id, KT_COMM_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) { kt_set_error("ncclCommInitRank: %s", g_rccl.GetErrorString(r)); delete c; return KT_ERR_HIP; }
    KT_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    KT_HIP(hipEventCreateWithFlags(&c->poses_ready, hipEventDisableTiming));
    *out = c;
    return KT_OK;
}

int kt_pose_gather(kt_comm* c, kt_tracker* t, int k, float* all_poses_host)
{
    KT_ARG(c && t && k > 0 && all_poses_host);
    const size_t n = (size_t)k * 16;
    if (c->cap_floats < n) {
        KT_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->send); (void)hipFree(c->recv);
        c->send = c->recv = nullptr; c->cap_floats = 0;
        KT_HIP(hipMalloc((void**)&c->send, n * sizeof(float)));
        KT_HIP(hipMalloc((void**)&c->recv, n * sizeof(float) * (size_t)c->nranks));
        c->cap_floats = n;
    }
    // the poses are written on the tracker's stream; the collective waits for exactly that copy, not for the frames behind it
    KT_TRY(kt_tracker_export_poses_device(t, k, c->send));
    KT_HIP(hipEventRecord(c->poses_ready, c->ctx->stream));
    KT_HIP(hipStreamWaitEvent(c->stream, c->poses_ready, 0));
    KT_NCCL(g_rccl.AllGather(c->send, c->recv, n, ncclFloat32, c->comm, c->stream));
    KT_HIP(hipMemcpyAsync(all_poses_host, c->recv, n * sizeof(float) * (size_t)c->nranks, hipMemcpyDeviceToHost, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
    return KT_OK;
}

int kt_comm_destroy(kt_comm* c)
{
    if (!c) return KT_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    (void)hipFree(c->send); (void)hipFree(c->recv);
    if (c->poses_ready) (void)hipEventDestroy(c->poses_ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return KT_OK;
}

}  // extern "C"
