#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void setflag(unsigned int* f, unsigned int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void spin(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64); }
__global__ void stamp(unsigned long long* out) { *out = __builtin_amdgcn_s_memrealtime(); }
int main() {
    int can = -1;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue %d\n", can);
    unsigned int* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("signal malloc: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { e = hipMalloc((void**)&sig, 8); printf("plain malloc: %s\n", hipGetErrorString(e)); }
    hipMemset(sig, 0, 8);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned long long* st; hipMalloc((void**)&st, 16);
    for (int rep = 0; rep < 5; ++rep) {
        // stream b waits for value rep+1, then stamps; stream a: spin ~50us, stamp, set flag
        e = hipStreamWaitValue32(b, sig, rep + 1, hipStreamWaitValueGte, 0xffffffff);
        if (e != hipSuccess) { printf("wait value: %s\n", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, b, st + 1);
        hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, a, 2000);
        hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, a, sig, (unsigned)(rep + 1));
        hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, a, st);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
        printf("rep %d: waiter stamped %.2f us after the setter's next kernel\n", rep, ((double)h[1] - (double)h[0]) / 100.0);
    }
    return 0;
}
