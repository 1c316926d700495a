// how fast can the host read a few KB a kernel has just written into pinned host memory, by allocation flavour?
//   hipcc --offload-arch=gfx950 -O2 -o pinned_read pinned_read.hip && ./pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void fill(uint4* p, int n, unsigned v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_uint4(v + i, v, v, v);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int bytes = 8192, n = bytes / 16;
    struct { const char* name; unsigned flags; } kinds[] = {{"default", hipHostMallocDefault}, {"noncoherent", hipHostMallocNonCoherent},
                                                          {"coherent", hipHostMallocCoherent}, {"mapped|portable", hipHostMallocMapped | hipHostMallocPortable},
                                                          {"numa-user", hipHostMallocNumaUser}};
    hipStream_t st;
    hipStreamCreate(&st);
    std::vector<char> local(bytes);
    for (auto& k : kinds) {
        void* h = nullptr;
        if (hipHostMalloc(&h, bytes, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); continue; }
        double best_sync = 1e9, best_copy = 1e9, best_sum = 1e9;
        for (int it = 0; it < 20; it++) {
            double t0 = now_us();
            hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, st, (uint4*)h, n, (unsigned)it);
            hipStreamSynchronize(st);
            double t1 = now_us();
            std::memcpy(local.data(), h, bytes);
            double t2 = now_us();
            unsigned long long s = 0;
            for (int i = 0; i < bytes / 8; i++) s += ((volatile unsigned long long*)h)[i];
            double t3 = now_us();
            if (((unsigned*)local.data())[0] != (unsigned)it) printf("stale!\n");
            best_sync = std::min(best_sync, t1 - t0), best_copy = std::min(best_copy, t2 - t1), best_sum = std::min(best_sum, t3 - t2);
            if (s == 42) printf("x");
        }
        printf("%-18s launch+sync %6.2f us   memcpy 8 KB out of it %6.2f us   second read (8-byte loads) %6.2f us\n", k.name, best_sync, best_copy, best_sum);
        hipHostFree(h);
    }
    // device buffer + hipMemcpyAsync D2H into pinned / pageable
    void* d; hipMalloc(&d, bytes);
    void* hp; hipHostMalloc(&hp, bytes, hipHostMallocDefault);
    for (int pageable = 0; pageable < 2; pageable++) {
        double best = 1e9, bestc = 1e9;
        for (int it = 0; it < 20; it++) {
            double t0 = now_us();
            hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, st, (uint4*)d, n, (unsigned)it);
            hipMemcpyAsync(pageable ? (void*)local.data() : hp, d, bytes, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
            double t1 = now_us();
            std::vector<char> l2(bytes);
            std::memcpy(l2.data(), pageable ? (void*)local.data() : hp, bytes);
            double t2 = now_us();
            best = std::min(best, t1 - t0), bestc = std::min(bestc, t2 - t1);
        }
        printf("device buffer + D2H into %s: launch+copy+sync %6.2f us, memcpy out %6.2f us\n", pageable ? "pageable" : "pinned  ", best, bestc);
    }
    return 0;
}
