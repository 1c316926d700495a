// fp.cuh -- BLS12-381 Fr / Fq Montgomery arithmetic for gfx950 (CDNA4), in registers.
//
// Representation: little-endian 32-bit limbs (8 for Fr, 12 for Fq) of the SAME Montgomery
// form the reference keeps in memory (ark-ff 0.4.2 MontBackend: R = 2^256 / 2^384, 4 / 6
// u64 limbs), so HBM buffers are byte-identical to a Rust `Vec<Fr>` / `[u64; 6]`.
//
// The multiplier is a finely-integrated product-scanning (FIPS) Montgomery multiplication
// built on one primitive: a 96-bit accumulate  acc += a*b  = v_mad_u64_u32 (64-bit
// accumulate, carry-out to VCC) + v_addc_co_u32 (carry into the third word).  This is
// carry-propagated integer VALU work: no MFMA (north star), no 64-bit multiplies (the
// hardware has none), no local arrays that could fall out of registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zk {

typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------
// field configurations (constants re-derived in tests/test_oracle_anchors.py)
// ---------------------------------------------------------------------------
struct FrCfg {
    static constexpr int N = 8;
    static constexpr u32 INV = 0xffffffffu;  // -r^{-1} mod 2^32
    __host__ __device__ static constexpr u32 P(int i) {
        constexpr u32 t[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                              0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 ONE(int i) {  // R mod r
        constexpr u32 t[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                              0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 R2(int i) {  // R^2 mod r
        constexpr u32 t[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                              0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return t[i];
    }
};

struct FqCfg {
    static constexpr int N = 12;
    static constexpr u32 INV = 0xfffcfffdu;  // -q^{-1} mod 2^32
    __host__ __device__ static constexpr u32 P(int i) {
        constexpr u32 t[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                               0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return t[i];
    }
    __host__ __device__ static constexpr u32 ONE(int i) {  // R mod q
        constexpr u32 t[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                               0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 R2(int i) {  // R^2 mod q
        constexpr u32 t[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                               0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return t[i];
    }
};

template <class C>
struct Fp {
    u32 l[C::N];
};
typedef Fp<FrCfg> Fr;
typedef Fp<FqCfg> Fq;

// ---------------------------------------------------------------------------
// carry primitives
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 addc(u32 a, u32 b, u32& carry) {
    u32 co;
    u32 r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
__device__ __forceinline__ u32 subb(u32 a, u32 b, u32& borrow) {
    u32 bo;
    u32 r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}

// (hi:lo) += a*b, 96-bit.  `b` may live in an SGPR (modulus limbs are wave-uniform).
__device__ __forceinline__ void mac3(u64& lo, u32& hi, u32 a, u32 b) {
#ifndef ZK_NO_ASM
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "v"(b)
        : "vcc");
#else
    u64 p = (u64)a * b;
    u64 s = lo + p;
    hi += (s < p) ? 1u : 0u;
    lo = s;
#endif
}
__device__ __forceinline__ void mac3s(u64& lo, u32& hi, u32 a, u32 b_uniform) {
#ifndef ZK_NO_ASM
    asm("v_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "s"(b_uniform)
        : "vcc");
#else
    mac3(lo, hi, a, b_uniform);
#endif
}
__device__ __forceinline__ void shift3(u64& lo, u32& hi) {
    lo = (lo >> 32) | ((u64)hi << 32);
    hi = 0;
}

// ---------------------------------------------------------------------------
// basic ops
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ Fp<C> fp_zero() {
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = 0;
    return r;
}
template <class C>
__device__ __forceinline__ Fp<C> fp_one() {
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = C::ONE(i);
    return r;
}
template <class C>
__device__ __forceinline__ bool fp_is_zero(const Fp<C>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) o |= a.l[i];
    return o == 0;
}
template <class C>
__device__ __forceinline__ bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// r = t - p if t >= p else t   (t < 2p)
template <class C>
__device__ __forceinline__ Fp<C> fp_reduce_once(const Fp<C>& t) {
    Fp<C> d, r;
    u32 bw = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) d.l[i] = subb(t.l[i], C::P(i), bw);
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = bw ? t.l[i] : d.l[i];
    return r;
}

template <class C>
__device__ __forceinline__ Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
    Fp<C> s;
    u32 c = 0;  // 2p < 2^(32N) for both fields: no carry out of the top limb
#pragma unroll
    for (int i = 0; i < C::N; i++) s.l[i] = addc(a.l[i], b.l[i], c);
    return fp_reduce_once<C>(s);
}

template <class C>
__device__ __forceinline__ Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
    Fp<C> d, r;
    u32 bw = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) d.l[i] = subb(a.l[i], b.l[i], bw);
    u32 mask = 0u - bw;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = addc(d.l[i], C::P(i) & mask, c);
    return r;
}

template <class C>
__device__ __forceinline__ Fp<C> fp_neg(const Fp<C>& a) {
    Fp<C> r;
    u32 bw = 0;
    bool z = fp_is_zero<C>(a);
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = subb(C::P(i), a.l[i], bw);
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = z ? 0u : r.l[i];
    return r;
}

template <class C>
__device__ __forceinline__ Fp<C> fp_dbl(const Fp<C>& a) {
    return fp_add<C>(a, a);
}

// ---------------------------------------------------------------------------
// Montgomery multiplication: r = a*b*R^{-1} mod p, FIPS.
// Invariant: the running value stays <= 2p-1 < 2^(32N) (both moduli leave >= 1 spare
// top bit), so the accumulator never needs more than the 3 words below and exactly one
// conditional subtraction canonicalises the result.
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ Fp<C> fp_mul(const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N;
    u32 m[N];
    Fp<C> t;
    u64 lo = 0;
    u32 hi = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int j = 0; j <= k; j++) mac3(lo, hi, a.l[j], b.l[k - j]);
#pragma unroll
        for (int j = 0; j < k; j++) mac3s(lo, hi, m[j], C::P(k - j));
        u32 mk = (u32)lo * C::INV;
        m[k] = mk;
        mac3s(lo, hi, mk, C::P(0));
        shift3(lo, hi);
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int j = k - N + 1; j < N; j++) mac3(lo, hi, a.l[j], b.l[k - j]);
#pragma unroll
        for (int j = k - N + 1; j < N; j++) mac3s(lo, hi, m[j], C::P(k - j));
        t.l[k - N] = (u32)lo;
        shift3(lo, hi);
    }
    t.l[N - 1] = (u32)lo;
    return fp_reduce_once<C>(t);
}

// Production multiplier: generated, one asm statement per accumulation run, hazard-safe carry
// rotation (tools/gen_fp_mul.py).  The generic template above is the readable reference and the
// -DZK_NO_ASM / -DZK_GENERIC_MUL fallback used to cross-check it.
#if !defined(ZK_NO_ASM) && !defined(ZK_GENERIC_MUL)
#include "fp_mul_gen.cuh"
#endif

template <class C>
__device__ __forceinline__ Fp<C> fp_sqr(const Fp<C>& a) {
    return fp_mul<C>(a, a);
}

// out of Montgomery form (ark-ff `into_bigint`): a * 1 * R^{-1}
template <class C>
__device__ __forceinline__ Fp<C> fp_from_mont(const Fp<C>& a) {
    Fp<C> one = fp_zero<C>();
    one.l[0] = 1;
    return fp_mul<C>(a, one);
}

// limb k of p - 2 (compile-time borrow chain; r ends in ...00000001 so the borrow propagates)
template <class C>
__host__ __device__ constexpr u32 fp_pm2_limb(int k) {
    u64 borrow = 2;
    u32 res = 0;
    for (int j = 0; j <= k; j++) {
        u64 pj = C::P(j);
        res = (u32)(pj - borrow);
        borrow = (pj < borrow) ? 1 : 0;
    }
    return res;
}

// a^(p-2): Fermat inverse (0 -> 0).  Exponent bits come from the compile-time modulus.
template <class C>
__device__ __noinline__ Fp<C> fp_inv(const Fp<C>& a) {
    Fp<C> acc = fp_one<C>();
    for (int i = C::N * 32 - 1; i >= 0; i--) {
        acc = fp_sqr<C>(acc);
        u32 w = 0;
#pragma unroll
        for (int k = 0; k < C::N; k++)
            if ((i >> 5) == k) w = fp_pm2_limb<C>(k);
        if ((w >> (i & 31)) & 1u) acc = fp_mul<C>(acc, a);
    }
    return acc;
}

// ---------------------------------------------------------------------------
// HBM <-> registers.  Fr = 32 B = two 16-byte accesses per lane; Fq = 48 B = three.
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ Fp<C> fp_load(const void* base, size_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + idx * (C::N * 4));
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) {
        uint4 v = p[i];
        r.l[4 * i] = v.x;
        r.l[4 * i + 1] = v.y;
        r.l[4 * i + 2] = v.z;
        r.l[4 * i + 3] = v.w;
    }
    return r;
}
template <class C>
__device__ __forceinline__ void fp_store(void* base, size_t idx, const Fp<C>& a) {
    uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + idx * (C::N * 4));
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) p[i] = make_uint4(a.l[4 * i], a.l[4 * i + 1], a.l[4 * i + 2], a.l[4 * i + 3]);
}

// convenient aliases
__device__ __forceinline__ Fr fr_load(const void* b, size_t i) { return fp_load<FrCfg>(b, i); }
__device__ __forceinline__ void fr_store(void* b, size_t i, const Fr& a) { fp_store<FrCfg>(b, i, a); }
__device__ __forceinline__ Fr fr_from_u4(const uint4& lo, const uint4& hi) {  // the two 16-byte halves of an element
    Fr r;
    r.l[0] = lo.x, r.l[1] = lo.y, r.l[2] = lo.z, r.l[3] = lo.w;
    r.l[4] = hi.x, r.l[5] = hi.y, r.l[6] = hi.z, r.l[7] = hi.w;
    return r;
}
__device__ __forceinline__ Fr fr_add(const Fr& a, const Fr& b) { return fp_add<FrCfg>(a, b); }
__device__ __forceinline__ Fr fr_sub(const Fr& a, const Fr& b) { return fp_sub<FrCfg>(a, b); }
__device__ __forceinline__ Fr fr_mul(const Fr& a, const Fr& b) { return fp_mul<FrCfg>(a, b); }
__device__ __forceinline__ Fq fq_add(const Fq& a, const Fq& b) { return fp_add<FqCfg>(a, b); }
__device__ __forceinline__ Fq fq_sub(const Fq& a, const Fq& b) { return fp_sub<FqCfg>(a, b); }
__device__ __forceinline__ Fq fq_mul(const Fq& a, const Fq& b) { return fp_mul<FqCfg>(a, b); }
__device__ __forceinline__ Fq fq_sqr(const Fq& a) { return fp_sqr<FqCfg>(a); }

}  // namespace zk
