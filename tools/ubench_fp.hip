// Micro-benchmark: Montgomery multiplication throughput (Fr, Fq) on gfx950 and the raw
// instruction rates it is bounded by.  Numbers feed DESIGN.md's integer-ALU roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../scalable-collaborative-zksnark_amd/csrc/fp.cuh"
using namespace zk;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class C, int ITER>
__global__ void __launch_bounds__(256) k_mul(u32* out, const u32* in) {
    Fp<C> x, y;
    size_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < C::N; i++) { x.l[i] = in[i] + (u32)tid; y.l[i] = in[C::N + i]; }
    x.l[C::N - 1] &= 0x0fffffff; y.l[C::N - 1] &= 0x0fffffff;
    for (int it = 0; it < ITER; it++) { x = fp_mul<C>(x, y); y = fp_mul<C>(y, x); }
    u32 s = 0;
    for (int i = 0; i < C::N; i++) s ^= x.l[i] ^ y.l[i];
    out[tid] = s;
}
template <class C, int ITER>
__global__ void __launch_bounds__(256) k_add(u32* out, const u32* in) {
    Fp<C> x, y;
    size_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < C::N; i++) { x.l[i] = in[i] + (u32)tid; y.l[i] = in[C::N + i]; }
    x.l[C::N - 1] &= 0x0fffffff; y.l[C::N - 1] &= 0x0fffffff;
    for (int it = 0; it < ITER; it++) { x = fp_add<C>(x, y); y = fp_sub<C>(y, x); }
    u32 s = 0;
    for (int i = 0; i < C::N; i++) s ^= x.l[i] ^ y.l[i];
    out[tid] = s;
}

constexpr int RI = 8192;
template <int MODE>
__global__ void __launch_bounds__(256) k_raw(u32* out, u32 seed) {
    u32 a = seed + threadIdx.x, b = seed * 3 + 1;
    u64 acc[8]; u32 w[8]; double d[8];
    for (int i = 0; i < 8; i++) { acc[i] = seed + i; w[i] = seed ^ i; d[i] = seed + i; }
    double da = 1.0000001, db = 0.9999999;
    for (int it = 0; it < RI; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            if (MODE == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            if (MODE == 2) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(w[i]) : "v"(a), "v"(b) : "vcc");
            if (MODE == 3) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(w[i]) : "v"(b) : "vcc");
            if (MODE == 4) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(w[i]) : "v"(b) : "vcc");
            if (MODE == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            if (MODE == 6) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            if (MODE == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"((u64)a));
            if (MODE == 8) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
            if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
            if (MODE == 10) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21");
            if (MODE == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i]) : "v"(b) : "vcc");
            if (MODE == 12) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(a), "v"(b));
            if (MODE == 13) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            if (MODE == 14) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            if (MODE == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(w[i]) : "v"(b));
        }
    }
    u64 s = 0; double ds = 0;
    for (int i = 0; i < 8; i++) { s += acc[i] + w[i]; ds += d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s + (u32)(s >> 32) + (u32)ds;
}

static double g_add_rate = 0;
template <class K>
int timeit(const char* name, K launch, double steps_per_launch, int instr) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CHECK(hipDeviceSynchronize());
    const int REP = 10;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < REP; i++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double per_s = steps_per_launch * REP / (ms * 1e-3);
    if (g_add_rate == 0) g_add_rate = per_s;
    printf("%-36s %9.3f ms/launch %10.2f G/s  x%5.2f of v_add_u32 time (%d instr)\n", name, ms / REP, per_s * 1e-9, g_add_rate / per_s, instr);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device CUs %d clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    u32 *d_out, *d_in;
    int maxblocks = p.multiProcessorCount * 8;
    CHECK(hipMalloc(&d_out, (size_t)maxblocks * 256 * 4));
    CHECK(hipMalloc(&d_in, 256));
    u32 h_in[64]; for (int i = 0; i < 64; i++) h_in[i] = 0x9e3779b9u * (i + 1);
    CHECK(hipMemcpy(d_in, h_in, 256, hipMemcpyHostToDevice));
    int blocks = maxblocks;
    double raw_steps = (double)blocks * 256 * RI * 8;
#define RAW(M, name, n) timeit(name, [&] { k_raw<M><<<blocks, 256>>>(d_out, 7); }, raw_steps, n)
    RAW(0, "v_add_u32", 1);
    RAW(15, "v_mov_b32", 1);
    RAW(12, "v_add3_u32", 1);
    RAW(3, "v_add_co_u32 (writes vcc)", 1);
    RAW(4, "v_addc_co_u32 (r/w vcc)", 1);
    RAW(11, "v_cndmask_b32 (reads vcc)", 1);
    RAW(1, "v_mad_u64_u32 (carry->vcc)", 1);
    RAW(10, "v_mad_u64_u32 (carry->s[20:21])", 1);
    RAW(2, "v_mad_u64_u32 + v_addc_co_u32", 2);
    RAW(5, "v_mul_lo_u32", 1);
    RAW(6, "v_mul_hi_u32", 1);
    RAW(7, "v_lshl_add_u64", 1);
    RAW(8, "v_fma_f64", 1);
    RAW(9, "v_mad_u32_u24", 1);
    RAW(13, "v_mul_u32_u24", 1);
    RAW(14, "v_mul_hi_u32_u24", 1);
    // field multiplications at several occupancies (blocks per CU of 256 threads = waves/SIMD)
    for (int bpc : {1, 2, 4, 8}) {
        int nb = p.multiProcessorCount * bpc;
        char nm[64];
        constexpr int IT = 512;
        snprintf(nm, 64, "Fr mul (%d waves/SIMD)", bpc);
        timeit(nm, [&] { k_mul<FrCfg, IT><<<nb, 256>>>(d_out, d_in); }, (double)nb * 256 * IT * 2, 136);
        snprintf(nm, 64, "Fq mul (%d waves/SIMD)", bpc);
        timeit(nm, [&] { k_mul<FqCfg, IT><<<nb, 256>>>(d_out, d_in); }, (double)nb * 256 * IT * 2, 300);
        snprintf(nm, 64, "Fq add/sub (%d waves/SIMD)", bpc);
        timeit(nm, [&] { k_add<FqCfg, IT * 4><<<nb, 256>>>(d_out, d_in); }, (double)nb * 256 * IT * 8, 36);
    }
    return 0;
}
