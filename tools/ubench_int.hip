// Micro-benchmark: issue rate of the integer instructions a Montgomery multiplier is
// built from, on gfx950.  Sets the integer-ALU roofline quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int CHAINS = 8;

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc[CHAINS];
    double dacc[CHAINS];
    for (int i = 0; i < CHAINS; i++) { acc[i] = seed + i; dacc[i] = (double)(seed + i); }
    double da = 1.0000001, db = 0.9999999;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (MODE == 0) {  // v_mad_u64_u32
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            } else if (MODE == 1) {  // v_mul_lo_u32
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (MODE == 2) {  // v_mul_hi_u32
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (MODE == 3) {  // v_add_co_u32 + v_addc_co_u32 (64-bit add, 2 instrs)
                uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                acc[i] = ((uint64_t)hi << 32) | lo;
            } else if (MODE == 4) {  // v_lshl_add_u64
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"((uint64_t)a));
            } else if (MODE == 5) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db));
            } else if (MODE == 6) {  // v_mad_u32_u24
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
                acc[i] = x;
            } else if (MODE == 7) {  // v_add_u32 (full-rate reference)
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (MODE == 8) {  // mad + addc (carry of the mad into a third word)
                uint32_t hi2 = (uint32_t)(dacc[i]);
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(hi2) : "v"(a), "v"(b) : "vcc");
                dacc[i] = hi2;
            } else if (MODE == 9) {  // v_mul_u32_u24 + v_mul_hi_u32_u24
                uint32_t x = (uint32_t)acc[i], y;
                asm volatile("v_mul_hi_u32_u24 %1, %0, %2\n\tv_mul_u32_u24 %0, %0, %2" : "+v"(x), "=&v"(y) : "v"(b));
                acc[i] = x ^ y;
            } else if (MODE == 10) {  // v_mad_i64_i32 
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            } else if (MODE == 11) {  // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dacc[i]) : "v"(da));
            } else if (MODE == 12) {  // v_pk_mul_lo_u16 
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (MODE == 13) {  // v_dot4_u32_u8 
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
                acc[i] = x;
            }
        }
    }
    uint64_t s = 0; double ds = 0;
    for (int i = 0; i < CHAINS; i++) { s += acc[i]; ds += dacc[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s + (uint32_t)(s >> 32) + (uint32_t)ds;
}

template <int MODE>
int run(const char* name, int instr_per_step, uint32_t* d_out, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(d_out, 7);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double steps = (double)blocks * 256 * ITER * CHAINS;
    double per_s = steps / (ms * 1e-3);
    // cycles per wave-step per SIMD at 2.4 GHz: 1024 SIMDs
    double cyc = 2.4e9 * 1024.0 * 64.0 / per_s;
    printf("%-34s %8.3f ms  %9.3f Gstep/s (lane)  ~%6.2f cyc/wave-step/SIMD (%d instr/step)\n", name, ms, per_s * 1e-9, cyc, instr_per_step);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, CUs %d, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    int blocks = p.multiProcessorCount * 8;  // 8 blocks x 4 waves = 8 waves/SIMD
    uint32_t* d_out; CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    run<7>("v_add_u32", 1, d_out, blocks);
    run<0>("v_mad_u64_u32", 1, d_out, blocks);
    run<10>("v_mad_i64_i32", 1, d_out, blocks);
    run<8>("v_mad_u64_u32 + v_addc_co_u32", 2, d_out, blocks);
    run<1>("v_mul_lo_u32", 1, d_out, blocks);
    run<2>("v_mul_hi_u32", 1, d_out, blocks);
    run<3>("v_add_co + v_addc_co", 2, d_out, blocks);
    run<4>("v_lshl_add_u64", 1, d_out, blocks);
    run<5>("v_fma_f64", 1, d_out, blocks);
    run<11>("v_mul_f64", 1, d_out, blocks);
    run<6>("v_mad_u32_u24", 1, d_out, blocks);
    run<9>("v_mul_hi_u32_u24 + v_mul_u32_u24", 2, d_out, blocks);
    run<12>("v_pk_mul_lo_u16", 1, d_out, blocks);
    run<13>("v_dot4_u32_u8", 1, d_out, blocks);
    return 0;
}
