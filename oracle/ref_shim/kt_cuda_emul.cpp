/*
 * kt_cuda_emul.cpp -- fiber scheduler + host-memory "device" allocator behind kt_cuda_emul.h.
 * TEST INFRASTRUCTURE ONLY (oracle/_ref).  One OS thread; every CUDA thread of the running block is a fiber.
 */
#include "kt_cuda_emul.h"

#include <vector>
#include <xmmintrin.h>
#include <pmmintrin.h>

#if !defined(__x86_64__)
#error "the fiber switch below is x86-64 System V only (the GPU boxes and the build container are x86-64)"
#endif

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

/* void ktemu_switch(void** save_sp, void* load_sp): callee-saved registers on the old stack, swap, restore, return */
extern "C" void ktemu_switch(void** save_sp, void* load_sp);
__asm__(".text\n"
        ".globl ktemu_switch\n"
        ".type ktemu_switch,@function\n"
        "ktemu_switch:\n"
        "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
        "  movq %rsp, (%rdi)\n"
        "  movq %rsi, %rsp\n"
        "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
        "  ret\n"
        ".size ktemu_switch,.-ktemu_switch\n");

namespace {

const size_t kStackBytes = 96 * 1024;

struct Fiber {
    void* sp;
    char* stack;
    bool done;
    int tid;
    uint3 tidx;
};

struct WarpState {
    int live, arrived;
    unsigned gen;
    unsigned slot[2][32];
    unsigned ballot[2];
};

std::vector<Fiber> g_fibers;          /* grows to the largest block seen; stacks are kept */
std::vector<WarpState> g_warps;
void* g_sched_sp;
Fiber* g_cur;
void (*g_body)(void*);
void* g_closure;
int g_live, g_bar_arrived;
unsigned g_bar_gen;
unsigned long long g_progress;

void die(const char* what)
{
    fprintf(stderr, "kt_cuda_emul: %s (block %u,%u,%u thread %u,%u,%u)\n", what, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x, threadIdx.y,
            threadIdx.z);
    abort();
}

void yield_to_scheduler()
{
    Fiber* me = g_cur;
    ktemu_switch(&me->sp, g_sched_sp);
    /* resumed: the scheduler has set g_cur / threadIdx */
}

void release_barrier_if_complete()
{
    if (g_bar_arrived > 0 && g_bar_arrived == g_live) {
        g_bar_arrived = 0;
        ++g_bar_gen;
        ++g_progress;
    }
}

void release_warp_if_complete(WarpState& w)
{
    if (w.arrived > 0 && w.arrived == w.live) {
        w.arrived = 0;
        ++w.gen;
        ++g_progress;
    }
}

void fiber_main()
{
    g_body(g_closure);
    Fiber* me = g_cur;
    me->done = true;
    --g_live;
    ++g_progress;
    WarpState& w = g_warps[me->tid >> 5];
    --w.live;
    /* a thread that has exited no longer takes part in barriers / collectives (pre-Volta behaviour) */
    release_barrier_if_complete();
    release_warp_if_complete(w);
    ktemu_switch(&me->sp, g_sched_sp);
    die("a finished fiber was resumed");
}

void prepare_fiber(Fiber& f)
{
    if (!f.stack) {
        f.stack = (char*)aligned_alloc(64, kStackBytes);
        if (!f.stack) die("out of memory for fiber stacks");
    }
    uintptr_t top = ((uintptr_t)(f.stack + kStackBytes)) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = 0;                       /* fake return address: fiber_main sees rsp % 16 == 8 like after a call */
    *--sp = (void*)&fiber_main;      /* ktemu_switch's ret lands here */
    for (int i = 0; i < 6; ++i) *--sp = 0;   /* rbp rbx r12 r13 r14 r15 */
    f.sp = (void*)sp;
    f.done = false;
}

void run_block(int nthreads)
{
    const int nwarps = (nthreads + 31) / 32;
    if ((int)g_fibers.size() < nthreads) g_fibers.resize(nthreads, Fiber{0, 0, true, 0, {0, 0, 0}});
    if ((int)g_warps.size() < nwarps) g_warps.resize(nwarps);
    for (int w = 0; w < nwarps; ++w) {
        g_warps[w].live = (w == nwarps - 1) ? nthreads - 32 * w : 32;
        g_warps[w].arrived = 0;
        g_warps[w].gen = 0;
    }
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = g_fibers[t];
        prepare_fiber(f);
        f.tid = t;
        f.tidx.x = t % blockDim.x;
        f.tidx.y = (t / blockDim.x) % blockDim.y;
        f.tidx.z = t / (blockDim.x * blockDim.y);
    }
    g_live = nthreads;
    g_bar_arrived = 0;
    g_bar_gen = 0;
    /* Schedule: any interleaving of the warps of a block is a legal execution.  This one lets warps 1.. run as far as they can before
     * warp 0 gets a turn, so that warp 0 is the last warp of its block to finish.  The reference's extractKernelSlice depends on that:
     * its thread 0 publishes output_count and zeroes global_count as soon as ITS warp leaves the z loop, with no block barrier in
     * front (extract.cu:289-303), so a warp of the last block that is still extracting then adds its points to the zeroed counter --
     * they overwrite the head of the output and are missing from the count.  (Found by tests/test_oracle_vs_ref.py::test_randomized_sweep
     * under a plain round-robin schedule; a race of the reference, not a property to reproduce.) */
    auto run_range = [&](int t0, int t1) {
        for (int t = t0; t < t1; ++t) {
            Fiber& f = g_fibers[t];
            if (f.done) continue;
            g_cur = &f;
            threadIdx = f.tidx;
            ktemu_switch(&g_sched_sp, f.sp);
        }
    };
    while (g_live > 0) {
        unsigned long long before = g_progress;
        for (;;) {   // warps 1 .. : until they are done or all of them wait for warp 0 (a block barrier)
            const unsigned long long b2 = g_progress;
            run_range(nthreads < 32 ? nthreads : 32, nthreads);
            if (g_progress == b2) break;
        }
        run_range(0, nthreads < 32 ? nthreads : 32);
        if (g_live > 0 && g_progress == before)
            die("deadlock: a barrier or warp collective was reached by only part of its threads");
    }
}

}  // namespace

namespace ktemu {

void launch_impl(dim3 grid, dim3 block, void (*thread_body)(void*), void* closure)
{
    if (g_cur) die("nested kernel launch");
    const unsigned long long nthreads = (unsigned long long)block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > 1024) die("invalid block size");
    /* the reference is built with --ftz=true (CMakeLists.txt:47): single-precision denormal inputs and results are zero
     * inside kernels.  SSE's FTZ + DAZ give exactly that for the scalar float code the kernels compile to. */
    const unsigned int mxcsr = _mm_getcsr();
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    g_body = thread_body;
    g_closure = closure;
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx;
                blockIdx.y = by;
                blockIdx.z = bz;
                run_block((int)nthreads);
            }
    g_cur = 0;
    _mm_setcsr(mxcsr);
}

void block_barrier()
{
    if (!g_cur) die("__syncthreads outside a kernel");
    unsigned gen = g_bar_gen;
    ++g_bar_arrived;
    release_barrier_if_complete();
    while (g_bar_gen == gen) {
        Fiber* me = g_cur;
        yield_to_scheduler();
        (void)me;
    }
}

unsigned lane_id() { return g_cur ? (unsigned)(g_cur->tid & 31) : 0u; }

/* Rendez-vous of the live lanes of the caller's warp.  Values are double-buffered by the parity of the warp's
 * generation: a lane can be at most one rendez-vous ahead of the slowest lane of its warp. */
static int warp_arrive(unsigned v, int pred)
{
    if (!g_cur) die("warp collective outside a kernel");
    WarpState& w = g_warps[g_cur->tid >> 5];
    const int lane = g_cur->tid & 31;
    const int p = (int)(w.gen & 1u);
    if (w.arrived == 0) w.ballot[p] = 0u;
    w.slot[p][lane] = v;
    if (pred) w.ballot[p] |= 1u << lane;
    const unsigned gen = w.gen;
    ++w.arrived;
    release_warp_if_complete(w);
    while (g_warps[g_cur->tid >> 5].gen == gen) yield_to_scheduler();
    return p;
}

unsigned warp_exchange(unsigned v, int src_lane)
{
    const int warp = g_cur ? (g_cur->tid >> 5) : 0;
    const int p = warp_arrive(v, 0);
    return g_warps[warp].slot[p][src_lane & 31];
}

unsigned warp_ballot(int pred)
{
    const int warp = g_cur ? (g_cur->tid >> 5) : 0;
    const int p = warp_arrive(0u, pred);
    return g_warps[warp].ballot[p];
}

}  // namespace ktemu

/* ---- "device" memory is host memory ---- */
static size_t g_pitch_align = 256;   /* rows of 2-D allocations are padded to this many bytes (exercises pitch handling) */
extern "C" void ktref_set_pitch_alignment(int bytes) { g_pitch_align = bytes > 0 ? (size_t)bytes : 1; }

/* Every "device" allocation sits between two guard zones that are checked when it is freed: a kernel of the reference that writes
 * outside its buffer (some do for argument combinations the reference itself never uses) is reported instead of corrupting the heap. */
static const size_t kGuard = 4096;
struct AllocHeader { size_t n; size_t magic; };
cudaError_t cudaMalloc(void** p, size_t n)
{
    const size_t body = (n + 255) & ~(size_t)255;
    char* raw = (char*)aligned_alloc(256, kGuard + body + kGuard);
    if (!raw) return cudaErrorMemoryAllocation;
    memset(raw, 0xA5, kGuard + body + kGuard);
    AllocHeader h = {n, 0x6b745f7265665f31ull};
    memcpy(raw, &h, sizeof(h));
    memset(raw + kGuard, 0xCD, n);   /* fresh device memory is not zero: make reads of it visible */
    *p = raw + kGuard;
    return cudaSuccess;
}

cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t width_bytes, size_t height)
{
    *pitch = (width_bytes + g_pitch_align - 1) / g_pitch_align * g_pitch_align;
    return cudaMalloc(p, *pitch * (height ? height : 1));
}

cudaError_t cudaFree(void* p)
{
    if (!p) return cudaSuccess;
    char* raw = (char*)p - kGuard;
    AllocHeader h;
    memcpy(&h, raw, sizeof(h));
    if (h.magic != 0x6b745f7265665f31ull) { fprintf(stderr, "kt_cuda_emul: cudaFree of a pointer cudaMalloc did not return\n"); abort(); }
    const size_t body = (h.n + 255) & ~(size_t)255;
    for (size_t i = sizeof(h); i < kGuard; ++i)
        if ((unsigned char)raw[i] != 0xA5) { fprintf(stderr, "kt_cuda_emul: a kernel wrote %zu bytes BELOW a device buffer of %zu bytes\n", kGuard - i, h.n); abort(); }
    for (size_t i = h.n; i < body + kGuard; ++i)
        if (i >= body ? (unsigned char)raw[kGuard + i] != 0xA5 : false) { fprintf(stderr, "kt_cuda_emul: a kernel wrote %zu bytes PAST a device buffer of %zu bytes\n", i - h.n + 1, h.n); abort(); }
    free(raw);
    return cudaSuccess;
}
cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind) { memmove(dst, src, n); return cudaSuccess; }
cudaError_t cudaMemset(void* p, int v, size_t n) { memset(p, v, n); return cudaSuccess; }

cudaError_t cudaMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t height, cudaMemcpyKind)
{
    for (size_t y = 0; y < height; ++y) memmove((char*)dst + y * dpitch, (const char*)src + y * spitch, width_bytes);
    return cudaSuccess;
}

/* containers/initialization.hpp:66 declares it, initialization.cpp (driver API, not compiled here) defines it */
void error(const char* error_string, const char* file, const int line, const char* func)
{
    fprintf(stderr, "Error: %s\t%s:%d %s\n", error_string, file, line, func ? func : "");
    exit(0);
}
