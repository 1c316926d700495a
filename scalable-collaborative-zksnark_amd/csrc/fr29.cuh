// fr29.cuh -- BLS12-381 Fr in UNSATURATED form for the multiplication-bound sumcheck passes: 9 limbs of 29
// bits in registers, Montgomery reduction by 2^261 (the scheme of fq30.cuh, see there for the rationale:
// v_mad_u64_u32 chains without a carry instruction, lazy reduction with static bounds).
//
// HBM data stays exactly the reference's: 4 x u64 Montgomery form with R = 2^256.  Values are re-limbed on load
// (no conversion multiplication); a product of two such values comes out as
//     f29_mul(x R, y R) = x y R^2 2^-261 = (x y R) 2^-5,
// so (1) multiplying by a CONSTANT c uses the pre-scaled constant 32 c (then the product is exact: the fold
// lo + r (hi - lo) of dsumcheck.rs:14-19 with r' = 32 r prepared on the host), and (2) sums of table products
// carry the common factor 2^-5, removed once on the handful of final sums (host, 5 modular doublings).
// r = 1 (mod 2^29): the Montgomery quotient digit is m = -acc mod 2^29 (no multiplication) and m * r_0 = m.
// 2^261 / r = 70.66: a product of inputs < A r and < B r is < (A B / 70.66 + 1) r; everything that is
// multiplied or stored is normalised (limbs 0..7 < 2^29); the top limb has 32 bits, values up to 565 r fit.
#pragma once
#include "fp.cuh"

namespace zk {

struct Fr29 {
    u32 l[9];  // value = sum l[i] * 2^(29 i)
};

struct R29 {
    static constexpr u32 MASK = 0x1fffffffu;
    // P = r, Pk = k r (normalised limbs); KRk = k r in redundant limbs (every limb >= any normalised limb it absorbs)
    __host__ __device__ static constexpr u32 P(int i) {
        constexpr u32 t[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 P2(int i) {
        constexpr u32 t[9] = {0x00000002u, 0x1ffffff0u, 0x1f2dff7fu, 0x16900bffu, 0x1b00aa77u, 0x180809a1u, 0x0a4199ceu, 0x14ca675fu, 0x00e7db4eu};
        return t[i];
    }
    __host__ __device__ static constexpr u32 P4(int i) {
        constexpr u32 t[9] = {0x00000004u, 0x1fffffe0u, 0x1e5bfeffu, 0x0d2017ffu, 0x160154efu, 0x10101343u, 0x1483339du, 0x0994cebeu, 0x01cfb69du};
        return t[i];
    }
    __host__ __device__ static constexpr u32 P8(int i) {
        constexpr u32 t[9] = {0x00000008u, 0x1fffffc0u, 0x1cb7fdffu, 0x1a402fffu, 0x0c02a9deu, 0x00202687u, 0x0906673bu, 0x13299d7du, 0x039f6d3au};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KR2(int i) {
        constexpr u32 t[9] = {0x20000002u, 0x3fffffefu, 0x3f2dff7eu, 0x36900bfeu, 0x3b00aa76u, 0x380809a0u, 0x2a4199cdu, 0x34ca675eu, 0x00e7db4du};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KR4(int i) {
        constexpr u32 t[9] = {0x20000004u, 0x3fffffdfu, 0x3e5bfefeu, 0x2d2017feu, 0x360154eeu, 0x30101342u, 0x3483339cu, 0x2994cebdu, 0x01cfb69cu};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KR8(int i) {
        constexpr u32 t[9] = {0x20000008u, 0x3fffffbfu, 0x3cb7fdfeu, 0x3a402ffeu, 0x2c02a9ddu, 0x20202686u, 0x2906673au, 0x33299d7cu, 0x039f6d39u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KR16(int i) {
        constexpr u32 t[9] = {0x20000010u, 0x3fffff7fu, 0x396ffbfeu, 0x34805ffeu, 0x380553bcu, 0x20404d0du, 0x320cce75u, 0x26533af9u, 0x073eda74u};
        return t[i];
    }
};

// 2^261 mod r in 29-bit limbs: f29_mul(x, 2^261) = x (mod r), a full reduction of a lazily grown value
#define ZK_R29_2P261 0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu

__device__ __forceinline__ Fr29 f29_zero() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
// carry normalisation: limbs 0..7 back below 2^29 (value unchanged); input limbs < 2^32 - 8
__device__ __forceinline__ void f29_norm(Fr29& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.l[i + 1] += a.l[i] >> 29;
        a.l[i] &= R29::MASK;
    }
}
__device__ __forceinline__ Fr29 f29_add(const Fr29& a, const Fr29& b) {  // bounds add
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    f29_norm(r);
    return r;
}
__device__ __forceinline__ Fr29 f29_add2x(const Fr29& a, const Fr29& b) {  // a + 2 b
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + 2u * b.l[i];
    f29_norm(r);
    return r;
}
// a + k r - b  for normalised b < k r
#define ZK_F29_SUB(name, KR)                                                      \
    __device__ __forceinline__ Fr29 name(const Fr29& a, const Fr29& b) {         \
        Fr29 r;                                                                    \
        _Pragma("unroll") for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + R29::KR(i) - b.l[i]; \
        f29_norm(r);                                                               \
        return r;                                                                  \
    }
ZK_F29_SUB(f29_sub2, KR2)
ZK_F29_SUB(f29_sub4, KR4)
ZK_F29_SUB(f29_sub8, KR8)
ZK_F29_SUB(f29_sub16, KR16)

// a * b * 2^-261 (mod r): normalised inputs < A r, < B r with A B <= 4000; result normalised, < (A B / 70.66 + 1) r.
// Product scanning; a column holds at most 9 + 9 limb products (< 2^63 with the carry): no mid-column folds.
__device__ __forceinline__ Fr29 f29_mul(const Fr29& a, const Fr29& b) {
    u32 m[9];
    Fr29 t;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) acc += (u64)a.l[i] * b.l[j];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 9) acc += (u64)m[i] * R29::P(j);
        }
        if (k < 9) {
            const u32 mk = (0u - (u32)acc) & R29::MASK;  // -r^-1 = -1 (mod 2^29)
            m[k] = mk;
            acc += mk;  // m_k * r_0, r_0 = 1: the low 29 bits are now zero
        } else {
            t.l[k - 9] = (u32)acc & R29::MASK;
        }
        acc >>= 29;
    }
    t.l[8] = (u32)acc;
    return t;
}

// v - c if v >= c else v, for a normalised constant c (limb accessor) and normalised v
#define ZK_F29_CSUB(name, C)                                                                         \
    __device__ __forceinline__ Fr29 name(const Fr29& v) {                                           \
        Fr29 d, r;                                                                                    \
        u32 bw = 0;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 9; i++) {                                              \
            u32 x = v.l[i] - R29::C(i) - bw;                                                          \
            bw = x >> 31;                                                                             \
            d.l[i] = (i < 8) ? (x & R29::MASK) : x;                                                   \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 9; i++) r.l[i] = bw ? v.l[i] : d.l[i];                \
        return r;                                                                                     \
    }
ZK_F29_CSUB(f29_csub_r, P)
ZK_F29_CSUB(f29_csub_2r, P2)
ZK_F29_CSUB(f29_csub_4r, P4)
ZK_F29_CSUB(f29_csub_8r, P8)
// canonical representative of a normalised value < 4 r / < 16 r
__device__ __forceinline__ Fr29 f29_canon4(const Fr29& v) { return f29_csub_r(f29_csub_2r(v)); }
__device__ __forceinline__ Fr29 f29_canon16(const Fr29& v) { return f29_csub_r(f29_csub_2r(f29_csub_4r(f29_csub_8r(v)))); }

// ---- 8 x 32-bit words (the Fr of fp.cuh / the 32-byte HBM element) <-> 9 x 29-bit limbs: pure re-limbing ----
__device__ __forceinline__ Fr29 f29_from_fr(const Fr& a) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, wi = bit >> 5, s = bit & 31;
        u32 v = a.l[wi] >> s;
        if (s > 3 && wi + 1 < 8) v |= a.l[wi + 1] << (32 - s);
        r.l[i] = (i < 8) ? (v & R29::MASK) : v;
    }
    return r;
}
__device__ __forceinline__ Fr f29_to_fr(const Fr29& a) {  // a normalised, < 2^256
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, li = bit / 29, s = bit % 29;  // word j starts inside limb li at offset s
        u32 v = a.l[li] >> s;
        if (li + 1 < 9) v |= a.l[li + 1] << (29 - s);
        if (s > 26 && li + 2 < 9) v |= a.l[li + 2] << (58 - s);
        r.l[j] = v;
    }
    return r;
}

}  // namespace zk
