// clock_mix.hip -- does the chip clock down under multiply-dense integer code?  Full-chip kernels with a fixed number of VALU
// instructions per lane and a varying share of v_mad_u64_u32 (the rest v_add_u32 / v_and_b32 on independent registers); for each mix:
// shader cycles per wave-instruction (s_memtime), effective shader clock (s_memtime ticks per 100 MHz wall tick) and wall time.
//   hipcc --offload-arch=gfx950 -O2 -o clock_mix clock_mix.hip && ./clock_mix [waves per SIMD = 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MADS, int ADDS>  // per group: MADS mads then ADDS cheap ops, all on independent registers
__global__ void __launch_bounds__(256) k_mix(unsigned long long* out, int iters, unsigned seed) {
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned x0 = seed * 3 + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned m0 = 0x1fffffffu ^ threadIdx.x, m1 = 0x12345679u + blockIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            if (MADS >= 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(m0), "v"(m1) : "vcc");
            if (MADS >= 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(m0), "v"(m1) : "vcc");
            if (MADS >= 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(m0), "v"(m1) : "vcc");
            if (MADS >= 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(m0), "v"(m1) : "vcc");
            if (MADS >= 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a4) : "v"(m0), "v"(m1) : "vcc");
            if (MADS >= 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a5) : "v"(m0), "v"(m1) : "vcc");
            if (MADS >= 7) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a6) : "v"(m0), "v"(m1) : "vcc");
            if (MADS >= 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a7) : "v"(m0), "v"(m1) : "vcc");
            if (ADDS >= 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(m0));
            if (ADDS >= 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x1) : "v"(m1));
            if (ADDS >= 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(m0));
            if (ADDS >= 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x3) : "v"(m1));
            if (ADDS >= 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x4) : "v"(m0));
            if (ADDS >= 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x5) : "v"(m1));
            if (ADDS >= 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x6) : "v"(m0));
            if (ADDS >= 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x7) : "v"(m1));
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    const unsigned long long s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = c1 - c0;
        out[3 * blockIdx.x + 1] = w1 - w0;
        out[3 * blockIdx.x + 2] = s;
    }
}

template <int MADS, int ADDS>
static void run(unsigned long long* d, int blocks, int wps) {
    const int iters = 20000;
    std::vector<unsigned long long> h(3 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<MADS, ADDS>), dim3(blocks), dim3(256), 0, 0, d, iters, 12345u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int b = 0; b < blocks; b++) cyc += (double)h[3 * b], wall += (double)h[3 * b + 1];
    cyc /= blocks, wall /= blocks;
    const double instr = (double)iters * 4 * (MADS + ADDS);  // per wave
    // a SIMD runs `wps` waves: cycles per instruction and SIMD = cycles of one wave / (instructions of one wave * wps)
    printf("mads %d adds %d : %8.3f ms  clock %.3f GHz  %.2f cycles per instr and SIMD  (%.2f T mad/s, %.2f T instr/s)\n", MADS, ADDS, ms,
           cyc / wall * 0.1, cyc / (instr * wps), (double)blocks * 256 * iters * 4 * MADS / (ms * 1e-3) / 1e12,
           (double)blocks * 256 * instr / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;  // waves per SIMD: blocks of 4 waves, 256 CUs x 4 SIMDs
    const int blocks = 256 * wps;
    unsigned long long* d;
    hipMalloc(&d, 3 * blocks * 8);
    printf("waves per SIMD %d (%d blocks of 256)\n", wps, blocks);
    run<8, 0>(d, blocks, wps);
    run<6, 2>(d, blocks, wps);
    run<4, 4>(d, blocks, wps);
    run<2, 6>(d, blocks, wps);
    run<0, 8>(d, blocks, wps);
    run<8, 8>(d, blocks, wps);
    run<4, 8>(d, blocks, wps);
    return 0;
}
