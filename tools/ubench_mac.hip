// A/B micro-benchmark of 96-bit multiply-accumulate sequences (hazard-safe variants) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int RI = 2048;
// 12 dependent macs into one accumulator (one column), repeated
template <int MODE>
__global__ void __launch_bounds__(256) k(u32* out, u32 seed) {
    u32 a[12], b[12];
    for (int i = 0; i < 12; i++) { a[i] = seed * (i + 3) + threadIdx.x; b[i] = seed * (i + 7) ^ threadIdx.x; }
    u64 lo = seed; u32 hi = 0;
    for (int it = 0; it < RI; it++) {
        if (MODE == 0) {  // current: mad ; addc   (no wait states)
#pragma unroll
            for (int i = 0; i < 12; i++)
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a[i]), "v"(b[i]) : "vcc");
        } else if (MODE == 1) {  // mad ; s_nop 1 ; addc
#pragma unroll
            for (int i = 0; i < 12; i++)
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a[i]), "v"(b[i]) : "vcc");
        } else if (MODE == 2) {  // whole column in one statement, carries rotated through 3 SGPR pairs, software-pipelined
            u64 c0, c1, c2;
            asm volatile(
                "v_mad_u64_u32 %0, %2, %5, %17, %0\n\t"
                "v_mad_u64_u32 %0, %3, %6, %18, %0\n\t"
                "v_mad_u64_u32 %0, %4, %7, %19, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
                "v_mad_u64_u32 %0, %2, %8, %20, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
                "v_mad_u64_u32 %0, %3, %9, %21, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
                "v_mad_u64_u32 %0, %4, %10, %22, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
                "v_mad_u64_u32 %0, %2, %11, %23, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
                "v_mad_u64_u32 %0, %3, %12, %24, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
                "v_mad_u64_u32 %0, %4, %13, %25, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
                "v_mad_u64_u32 %0, %2, %14, %26, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
                "v_mad_u64_u32 %0, %3, %15, %27, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
                "v_mad_u64_u32 %0, %4, %16, %28, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
                "s_nop 0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
                "s_nop 1\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, %4"
                : "+v"(lo), "+v"(hi), "=&s"(c0), "=&s"(c1), "=&s"(c2)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
                  "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11])
                : "vcc");
        } else if (MODE == 3) {  // same as 0 but one statement for the column (removes the compiler's per-statement pad)
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %2, %14, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %3, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %16, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %17, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %18, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %19, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %8, %20, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %9, %21, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %10, %22, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %11, %23, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %12, %24, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %13, %25, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                : "+v"(lo), "+v"(hi)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
                  "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11])
                : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)lo ^ (u32)(lo >> 32) ^ hi;
}
template <int MODE> int run(const char* name, u32* d_out, int blocks, u32* h_chk) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) k<MODE><<<blocks, 256>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) k<MODE><<<blocks, 256>>>(d_out, 7);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h_chk, d_out, 4, hipMemcpyDeviceToHost));
    double macs = (double)blocks * 256 * RI * 12 * 10;
    printf("%-44s blocks/CU=%d  %8.3f ms  %8.2f Gmac/s  chk=%08x\n", name, blocks / 256, ms / 10, macs / (ms * 1e-3) * 1e-9, *h_chk);
    return 0;
}
int main() {
    u32* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4)); u32 chk;
    for (int bpc : {1, 2, 8}) {
        int blocks = 256 * bpc;
        run<0>("mad;addc (current, no wait states)", d_out, blocks, &chk);
        run<3>("mad;addc, one asm per column", d_out, blocks, &chk);
        run<1>("mad;s_nop 1;addc", d_out, blocks, &chk);
        run<2>("column, 3 rotating carry SGPRs, pipelined", d_out, blocks, &chk);
    }
    return 0;
}
