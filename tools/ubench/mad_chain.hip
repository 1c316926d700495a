// mad_chain.hip -- issue cost of v_mad_u64_u32 streams at low occupancy.  ONE asm statement per loop body (hipcc pads every asm
// statement boundary with s_nop, so a statement per instruction measures the pads), carry-out to an SGPR pair nobody reads.
//   modes: 8 mads on 8 accumulators | 8 mads on ONE accumulator (dependent chain) | 8 x (mad, add) | 8 x (mad, add, and) | 16 cheap ops
//   hipcc --offload-arch=gfx950 -O2 -o mad_chain mad_chain.hip && ./mad_chain [waves per SIMD = 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define M(i) "v_mad_u64_u32 %" #i ", s[20:21], %10, %11, %" #i "\n\t"
#define MD "v_mad_u64_u32 %0, s[20:21], %10, %11, %0\n\t"
#define A "v_add_u32 %8, %8, %10\n\t"
#define B "v_and_b32 %9, %9, %11\n\t"
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(x0), "+v"(x1) : "v"(m0), "v"(m1) : "s20", "s21"

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(unsigned long long* out, int iters, unsigned seed) {
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned x0 = seed * 3 + threadIdx.x, x1 = x0 + 1;
    const unsigned m0 = 0x1fffffffu ^ threadIdx.x, m1 = 0x12345679u + blockIdx.x;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            if (MODE == 0) asm volatile(M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) : OPS);
            if (MODE == 1) asm volatile(MD MD MD MD MD MD MD MD : OPS);
            if (MODE == 2) asm volatile(M(0) A M(1) B M(2) A M(3) B M(4) A M(5) B M(6) A M(7) B : OPS);
            if (MODE == 3) asm volatile(M(0) A B M(1) A B M(2) A B M(3) A B M(4) A B M(5) A B M(6) A B M(7) A B : OPS);
            if (MODE == 4) asm volatile(A B A B A B A B A B A B A B A B : OPS);
            if (MODE == 5) asm volatile(MD A MD B MD A MD B MD A MD B MD A MD B : OPS);
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ x0 ^ x1;
    if (threadIdx.x == 0) out[2 * blockIdx.x] = c1 - c0, out[2 * blockIdx.x + 1] = s;
}
template <int MODE>
static void run(const char* name, int mads, int cheap, unsigned long long* d, int blocks, int wps) {
    const int iters = 8000;
    std::vector<unsigned long long> h(2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters, 12345u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; b++) cyc += (double)h[2 * b];
    cyc /= blocks;
    const double groups = iters * 4.0;  // one asm statement = `mads` mads + `cheap` cheap ops
    printf("%-34s %7.3f ms  %7.2f cycles per statement and wave (%d mad + %d cheap)  %6.2f T mad/s  %6.2f T instr/s\n", name, ms, cyc / groups, mads, cheap,
           (double)blocks * 256 * groups * mads / (ms * 1e-3) / 1e12, (double)blocks * 256 * groups * (mads + cheap) / (ms * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int blocks = 256 * wps;
    unsigned long long* d;
    hipMalloc(&d, 2 * blocks * 8);
    printf("waves per SIMD %d\n", wps);
    run<0>("8 mads, 8 accumulators", 8, 0, d, blocks, wps);
    run<1>("8 mads, one accumulator", 8, 0, d, blocks, wps);
    run<2>("8 x (mad, cheap)", 8, 8, d, blocks, wps);
    run<5>("8 x (dependent mad, cheap)", 8, 8, d, blocks, wps);
    run<3>("8 x (mad, cheap, cheap)", 8, 16, d, blocks, wps);
    run<4>("16 cheap", 0, 16, d, blocks, wps);
    return 0;
}
