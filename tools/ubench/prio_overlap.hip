// Can a chain of small dependent kernels make progress WHILE a long kernel that fills every wave slot is running?
// long kernel: 1792 workgroups x 256 threads, 256 VGPRs (2 waves / SIMD like zk::k_accum_tiles), ~0.6 ms per workgroup;
// chain: 25 dependent launches of a small kernel (512 workgroups, ~8 us each alone) on a second stream, created with
// normal or HIGH priority.  Prints the chain's wall time alone, beside the long kernel at equal priority, and at high priority.
//   hipcc --offload-arch=gfx950 -O3 -o prio_overlap prio_overlap.hip && ./prio_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_long(unsigned* out, int iters) {
    // 200+ live registers: an array the compiler cannot shrink, mixed through a dependent multiply chain
    unsigned r[200];
#pragma unroll
    for (int i = 0; i < 200; i++) r[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 200; i++) r[i] = r[i] * 1664525u + r[(i + 7) % 200];
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 200; i++) s ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_small(unsigned* buf, int iters) {
    unsigned x = buf[blockIdx.x * 256 + threadIdx.x];
    for (int it = 0; it < iters; it++) x = x * 1664525u + 1013904223u;
    buf[blockIdx.x * 256 + threadIdx.x] = x;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    unsigned *d_long, *d_small;
    CHECK(hipMalloc(&d_long, 1792 * 256 * 4));
    CHECK(hipMalloc(&d_small, 512 * 256 * 4));
    CHECK(hipMemset(d_small, 1, 512 * 256 * 4));
    int lo, hi;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least, hi = greatest (numerically lower)
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    hipStream_t s_long, s_norm, s_high;
    CHECK(hipStreamCreateWithPriority(&s_long, hipStreamNonBlocking, lo));
    CHECK(hipStreamCreateWithPriority(&s_norm, hipStreamNonBlocking, lo));
    CHECK(hipStreamCreateWithPriority(&s_high, hipStreamNonBlocking, hi));
    const int LONG_IT = 2200, SMALL_IT = 3000, CHAIN = 25;
    // calibrate
    for (int w = 0; w < 2; w++) { k_long<<<1792, 256, 0, s_long>>>(d_long, LONG_IT); CHECK(hipStreamSynchronize(s_long)); }
    double t0 = now();
    k_long<<<1792, 256, 0, s_long>>>(d_long, LONG_IT);
    CHECK(hipStreamSynchronize(s_long));
    const double t_long = now() - t0;
    auto chain = [&](hipStream_t s) { for (int i = 0; i < CHAIN; i++) k_small<<<512, 256, 0, s>>>(d_small, SMALL_IT); };
    chain(s_norm); CHECK(hipStreamSynchronize(s_norm));
    t0 = now(); chain(s_norm); CHECK(hipStreamSynchronize(s_norm));
    const double t_chain = now() - t0;
    printf("alone: long kernel %.3f ms, chain of %d small kernels %.3f ms\n", t_long, CHAIN, t_chain);
    for (int mode = 0; mode < 2; mode++) {
        hipStream_t sc = mode ? s_high : s_norm;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipDeviceSynchronize());
            t0 = now();
            k_long<<<1792, 256, 0, s_long>>>(d_long, LONG_IT);
            chain(sc);
            CHECK(hipStreamSynchronize(sc));
            const double tc = now() - t0;
            CHECK(hipStreamSynchronize(s_long));
            const double tl = now() - t0;
            printf("%s priority chain beside the long kernel: chain done after %.3f ms, long kernel after %.3f ms (serial would be %.3f)\n", mode ? "HIGH  " : "equal ", tc, tl,
                   t_long + t_chain);
        }
    }
    return 0;
}
