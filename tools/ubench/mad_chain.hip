// mad_chain.hip -- issue cost of v_mad_u64_u32 accumulation chains: C independent 64-bit accumulators, each a chain of dependent
// mads (acc += a * b), optionally with a cheap independent instruction after every mad.  Reports cycles per mad and SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o mad_chain mad_chain.hip && ./mad_chain [waves per SIMD = 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CHAINS, int FILL>
__global__ void __launch_bounds__(256) k_chain(unsigned long long* out, int iters, unsigned seed) {
    unsigned long long a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * (2 * i + 3);
    unsigned x0 = seed * 3 + threadIdx.x, x1 = x0 + 1;
    const unsigned m0 = 0x1fffffffu ^ threadIdx.x, m1 = 0x12345679u + blockIdx.x;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 32; rep++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[rep % CHAINS]) : "v"(m0), "v"(m1) : "vcc");
            if (FILL >= 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(m0));
            if (FILL >= 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x1) : "v"(m1));
        }
    }
    const unsigned long long c1 = clock64();
    unsigned long long s = x0 ^ x1;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (threadIdx.x == 0) out[2 * blockIdx.x] = c1 - c0, out[2 * blockIdx.x + 1] = s;
}
template <int CHAINS, int FILL>
static void run(unsigned long long* d, int blocks, int wps) {
    const int iters = 4000;
    std::vector<unsigned long long> h(2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<CHAINS, FILL>), dim3(blocks), dim3(256), 0, 0, d, iters, 12345u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; b++) cyc += (double)h[2 * b];
    cyc /= blocks;
    printf("chains %d fill %d : %7.3f ms  %6.2f cycles per mad and wave, %5.2f per mad and SIMD   %.2f T mad/s\n", CHAINS, FILL, ms, cyc / (iters * 32.0),
           cyc / (iters * 32.0 * wps), (double)blocks * 256 * iters * 32 / (ms * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int blocks = 256 * wps;
    unsigned long long* d;
    hipMalloc(&d, 2 * blocks * 8);
    printf("waves per SIMD %d\n", wps);
    run<1, 0>(d, blocks, wps);
    run<2, 0>(d, blocks, wps);
    run<4, 0>(d, blocks, wps);
    run<8, 0>(d, blocks, wps);
    run<1, 1>(d, blocks, wps);
    run<1, 2>(d, blocks, wps);
    run<2, 1>(d, blocks, wps);
    run<2, 2>(d, blocks, wps);
    return 0;
}
