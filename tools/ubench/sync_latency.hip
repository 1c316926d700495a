// launch + hipStreamSynchronize round trip of an empty kernel under the device scheduling flags
//   hipcc --offload-arch=gfx950 -O2 -o sync_latency sync_latency.hip && ./sync_latency [0|1|2|4]   (auto, spin, yield, blocking)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void nop(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void flagk(volatile int* flag, int seq) {
    if (threadIdx.x == 0) {
        __atomic_store_n((int*)flag, seq, __ATOMIC_RELEASE);  // system-scope visibility for fine-grained host memory
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    unsigned flags = argc > 1 ? (unsigned)atoi(argv[1]) : 0;
    hipError_t e = hipSetDeviceFlags(flags);
    hipStream_t st;
    hipStreamCreate(&st);
    for (int i = 0; i < 10; i++) { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, st, (int*)nullptr); hipStreamSynchronize(st); }
    double best = 1e9, sum = 0;
    const int R = 200;
    for (int i = 0; i < R; i++) {
        double t0 = now_us();
        hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipStreamSynchronize(st);
        double d = now_us() - t0;
        best = d < best ? d : best; sum += d;
    }
    double best3 = 1e9;
    for (int i = 0; i < R; i++) {
        double t0 = now_us();
        for (int k = 0; k < 3; k++) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipStreamSynchronize(st);
        double d = now_us() - t0;
        best3 = d < best3 ? d : best3;
    }
    int* hflag = nullptr;
    hipHostMalloc((void**)&hflag, 64, hipHostMallocDefault);
    *hflag = 0;
    double bestf = 1e9;
    for (int i = 1; i <= R; i++) {
        double t0 = now_us();
        hipLaunchKernelGGL(flagk, dim3(1), dim3(64), 0, st, (volatile int*)hflag, i);
        while (*(volatile int*)hflag != i) { }
        double d = now_us() - t0;
        bestf = d < bestf ? d : bestf;
    }
    hipStreamSynchronize(st);
    printf("launch + spin on a pinned flag written by the kernel: min %.2f us\n", bestf);
    printf("flags %u (set: %s): launch+sync min %.2f us avg %.2f us; 3 launches + sync min %.2f us\n", flags, hipGetErrorString(e), best, sum / R, best3);
    return 0;
}
