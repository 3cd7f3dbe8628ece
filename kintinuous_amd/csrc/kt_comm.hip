// kt_comm.hip -- the one collective of the multi-stream layout: every rank (one tracker per GPU, one process per
// rank) contributes its last k dense poses -- k x 16 floats, row-major [R | currentGlobalCamera], KintinuousTracker.h:151-169 -- to
// ONE ncclAllGather over RCCL / xGMI, on a dedicated stream so it never queues behind a frame.  64 B - 2 KB messages: latency
// bound, no ring all-reduce, no data-path collective anywhere else.
// librccl is opened lazily (dlopen) so that libkt_hip.so loads -- and BUILDS -- on hosts without it: the five entry points used are
// declared here with the types of RCCL's public C API (nccl.h: ncclUniqueId = 128 opaque bytes, ncclResult_t / ncclDataType_t
// enums passed as int, ncclComm_t an opaque pointer), so rccl.h is not a build dependency.
#include "kt_internal.hpp"

#include <dlfcn.h>
#include <string.h>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;     // ncclSuccess = 0
typedef int ncclDataType_t;   // ncclFloat32 = 7
}
enum { ncclSuccess = 0, ncclFloat32 = 7, NCCL_UNIQUE_ID_BYTES = 128 };
typedef ncclResult_t (*kt_ncclGetUniqueId_t)(ncclUniqueId*);
typedef ncclResult_t (*kt_ncclCommInitRank_t)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*kt_ncclAllGather_t)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*kt_ncclCommDestroy_t)(ncclComm_t);
typedef const char* (*kt_ncclGetErrorString_t)(ncclResult_t);

struct kt_comm {
    kt_ctx* ctx;
    ncclComm_t comm;
    hipStream_t stream;
    hipEvent_t poses_ready;
    int rank, nranks;
    float* send; float* recv;   // device staging, grown on demand
    size_t cap_floats;
    float* token;               // [1 + nranks] floats for kt_comm_barrier
};

namespace {
struct Rccl {
    void* lib;
    kt_ncclGetUniqueId_t GetUniqueId;
    kt_ncclCommInitRank_t CommInitRank;
    kt_ncclAllGather_t AllGather;
    kt_ncclCommDestroy_t CommDestroy;
    kt_ncclGetErrorString_t GetErrorString;
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

int kt_comm_destroy(kt_comm* c);

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
    // (a failure from here on releases what has been built: the communicator, the stream)
    int s = kt_check(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreateWithFlags", __FILE__, __LINE__);
    if (s == KT_OK) s = kt_check(hipEventCreateWithFlags(&c->poses_ready, hipEventDisableTiming), "hipEventCreateWithFlags", __FILE__, __LINE__);
    if (s == KT_OK) s = kt_check(hipMalloc((void**)&c->token, 2 * sizeof(float) * (size_t)nranks + sizeof(float)), "hipMalloc", __FILE__, __LINE__);
    if (s != KT_OK) { (void)kt_comm_destroy(c); return s; }
    *out = c;
    return KT_OK;
}

// every rank has reached this call: a one-float all-gather on the communicator's stream (bench.py brackets its timed region with it,
// so a multi-GPU run needs no second communicator for barriers)
int kt_comm_barrier(kt_comm* c)
{
    KT_ARG(c);
    KT_NCCL(g_rccl.AllGather(c->token, c->token + 1, 1, ncclFloat32, c->comm, c->stream));
    KT_HIP(hipStreamSynchronize(c->stream));
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
    (void)hipFree(c->send); (void)hipFree(c->recv); (void)hipFree(c->token);
    if (c->poses_ready) (void)hipEventDestroy(c->poses_ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return KT_OK;
}

}  // extern "C"
