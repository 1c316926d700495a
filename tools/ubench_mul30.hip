// Experiment: unsaturated (13 x 30-bit limb) Montgomery multiplication for Fq vs the production
// saturated 12 x 32-bit FIPS multiplier.  Columns accumulate in 64 bits without any carry
// instruction; correctness is checked against the production multiplier through the map
// x -> x * 2^6 (R' = 2^390 vs R = 2^384).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../scalable-collaborative-zksnark_amd/csrc/fq30.cuh"  // production multiplier / squaring (includes fp.cuh)
using namespace zk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct F30 { u32 l[13]; };
__host__ __device__ constexpr u32 QL30(int i) {
    constexpr u32 t[13] = {0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673, 0x12e13ce1,
                           0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111};
    return t[i];
}
constexpr u32 QP30 = 0x3ffcfffd;
constexpr u32 MASK30 = 0x3fffffffu;

// r = a*b*2^-390 mod q, inputs < q with 30-bit limbs, output < q
__device__ __forceinline__ F30 mul30(const F30& a, const F30& b) {
    u32 m[13];
    F30 t;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        int cnt = 0;
        u64 nxt = 0;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            int j = k - i;
            if (j >= 0 && j < 13) { acc += (u64)a.l[i] * b.l[j]; cnt++; }
        }
#pragma unroll
        for (int i = 0; i < 13; i++) {
            int j = k - i;
            if (i < k && i < 13 && j >= 1 && j < 13) {
                if (cnt == 15) { nxt += acc >> 30; acc &= MASK30; cnt = 0; }
                acc += (u64)m[i] * QL30(j);
                cnt++;
            }
        }
        if (k < 13) {
            if (cnt >= 15) { nxt += acc >> 30; acc &= MASK30; }
            u32 mk = ((u32)acc * QP30) & MASK30;
            m[k] = mk;
            acc += (u64)mk * QL30(0);
        } else {
            t.l[k - 13] = (u32)acc & MASK30;
        }
        acc = (acc >> 30) + nxt;
    }
    t.l[12] = (u32)acc;
    // conditional subtraction of q
    F30 d; u32 bw = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) { u32 x = t.l[i] - QL30(i) - bw; bw = x >> 31; d.l[i] = x & MASK30; }
    // (top limb: 21 bits, the subtraction borrow shows in bit 31 too)
    F30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = bw ? t.l[i] : d.l[i];
    return r;
}

template <int ITER> __global__ void __launch_bounds__(256) k30(u32* out, const u32* in) {
    F30 x, y; size_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 13; i++) { x.l[i] = (in[i] + (u32)tid) & MASK30; y.l[i] = in[13 + i] & MASK30; }
    x.l[12] &= 0xfffff; y.l[12] &= 0xfffff;
    for (int it = 0; it < ITER; it++) { x = mul30(x, y); y = mul30(y, x); }
    u32 s = 0; for (int i = 0; i < 13; i++) s ^= x.l[i] ^ y.l[i];
    out[tid] = s;
}
template <int ITER> __global__ void __launch_bounds__(256) k32(u32* out, const u32* in) {
    Fq x, y; size_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 12; i++) { x.l[i] = in[i] + (u32)tid; y.l[i] = in[12 + i]; }
    x.l[11] &= 0x0fffffff; y.l[11] &= 0x0fffffff;
    for (int it = 0; it < ITER; it++) { x = fq_mul(x, y); y = fq_mul(y, x); }
    u32 s = 0; for (int i = 0; i < 12; i++) s ^= x.l[i] ^ y.l[i];
    out[tid] = s;
}
// the production routines of csrc/fq30.cuh (lazy reduction: no final conditional subtraction)
template <int ITER, bool SQR> __global__ void __launch_bounds__(256) kprod(u32* out, const u32* in) {
    Fq30 x, y; size_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 13; i++) { x.l[i] = (in[i] + (u32)tid) & MASK30; y.l[i] = (in[13 + i] ^ (u32)tid) & MASK30; }  // both lane-dependent
    x.l[12] &= 0xfffff; y.l[12] &= 0xfffff;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
        if (SQR) { x = f30_sqr(x); y = f30_sqr(y); }
        else { x = f30_mul(x, y); y = f30_mul(y, x); }
    }
    u32 s = 0; for (int i = 0; i < 13; i++) s ^= x.l[i] ^ y.l[i];
    out[tid] = s;
}
// correctness: c32 = fq_mul(a,b) (R=2^384); c30 = mul30(a,b) on the same integers: c30 * 2^390 == c32 * 2^384 (mod q)
// i.e. c32 == c30 * 2^6.  Checked on the host with __int128-free big arithmetic via repeated doubling on device.
__global__ void kcheck(const u32* in, u32* bad, int n) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x; if (tid >= n) return;
    Fq a, b; for (int i = 0; i < 12; i++) { a.l[i] = in[(tid * 24 + i)]; b.l[i] = in[tid * 24 + 12 + i]; }
    a.l[11] &= 0x0fffffff; b.l[11] &= 0x0fffffff;  // < 2^380 < q
    auto to30 = [](const Fq& x) { F30 r; for (int i = 0; i < 13; i++) { int bit = 30 * i, w = bit / 32, s = bit % 32; u64 v = x.l[w]; if (w + 1 < 12) v |= (u64)x.l[w + 1] << 32; r.l[i] = (u32)(v >> s) & MASK30; } return r; };
    auto to32 = [](const F30& x) { Fq r; for (int i = 0; i < 12; i++) r.l[i] = 0; for (int i = 0; i < 13; i++) { int bit = 30 * i, w = bit / 32, s = bit % 32; u64 v = (u64)x.l[i] << s; r.l[w] |= (u32)v; if (w + 1 < 12) r.l[w + 1] |= (u32)(v >> 32); } return r; };
    Fq c32 = fq_mul(a, b);
    Fq c30 = to32(mul30(to30(a), to30(b)));
    for (int k = 0; k < 6; k++) c30 = fq_add(c30, c30);
    bool ok = fp_eq<FqCfg>(c32, c30);
    if (!ok) atomicAdd(bad, 1);
}
template <class K> int timeit(const char* name, K launch, double muls) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) launch();
    CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; i++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %8.3f ms  %7.2f G mul/s\n", name, ms / 5, muls * 5 / (ms * 1e-3) * 1e-9); return 0;
}
int main() {
    u32 *d_out, *d_in, *d_bad; const int NCHK = 1 << 16;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4)); CHECK(hipMalloc(&d_in, NCHK * 24 * 4)); CHECK(hipMalloc(&d_bad, 4));
    u32* h = new u32[NCHK * 24]; u64 s = 88172645463325252ULL;
    for (int i = 0; i < NCHK * 24; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (u32)(s >> 16); }
    CHECK(hipMemcpy(d_in, h, NCHK * 24 * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(d_bad, 0, 4));
    kcheck<<<NCHK / 256, 256>>>(d_in, d_bad, NCHK); u32 bad; CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("correctness: %u mismatches of %d\n", bad, NCHK);
    constexpr int IT = 256;
    for (int bpc : {1, 2, 4, 8}) {
        int nb = 256 * bpc; char nm[64];
        snprintf(nm, 64, "Fq mul 12x32 FIPS (%d waves/SIMD)", bpc);
        timeit(nm, [&] { k32<IT><<<nb, 256>>>(d_out, d_in); }, (double)nb * 256 * IT * 2);
        snprintf(nm, 64, "Fq mul 13x30 unsaturated (%d waves/SIMD)", bpc);
        timeit(nm, [&] { k30<IT><<<nb, 256>>>(d_out, d_in); }, (double)nb * 256 * IT * 2);
        snprintf(nm, 64, "fq30.cuh f30_mul (%d waves/SIMD)", bpc);
        timeit(nm, [&] { kprod<IT, false><<<nb, 256>>>(d_out, d_in); }, (double)nb * 256 * IT * 2);
        snprintf(nm, 64, "fq30.cuh f30_sqr (%d waves/SIMD)", bpc);
        timeit(nm, [&] { kprod<IT, true><<<nb, 256>>>(d_out, d_in); }, (double)nb * 256 * IT * 2);
    }
    return 0;
}
