// EXPERIMENT RECORD (round 3, not part of the build): the field arithmetic of the product-sumcheck passes on 9 x 29-bit digits.
// Bit-exact (62 GPU tests), NOT faster: profiles/r03l_fr29_and_issue_model.txt.  The kernel that used it (k_pass29) was removed again.
// fr29.cuh -- BLS12-381 Fr for the product-sumcheck HBM passes in UNSATURATED form: 9 digits of 29 bits in registers,
// Montgomery radix R' = 2^261 (the same idea as fq30.cuh for Fq).
//
// Why: on gfx950 v_mad_u64_u32 AND v_addc_co_u32 both issue in 4 cycles, plain 32-bit add / and / shift / move in 2
// (profiles/r01_ubench_int_alu.txt).  The saturated 8 x 32-bit multiplier of fp.cuh pays one v_addc_co_u32 per limb product
// (128 mad + 128 addc per multiplication, 64 + 64 per wide accumulation): the counters of k_pass<2, 1> show a kernel that
// issues in 96 % of its VALU slots with only 38 % of them multiplies (profiles/r03k_sc_valu_product.csv).  With 29-bit digits
// a digit product is < 2^58, so the 9 + 9 products of a column add up in ONE 64-bit register with no carry instruction at
// all: a Montgomery multiplication is 153 mad + ~60 two-cycle instructions, a wide accumulation 81 mad + ~50.
//
// Lazy reduction: 2^261 / r = 70.6, so values up to 70 r fit and the hot path has no conditional subtraction; a value is
// brought back below r only when it is stored (two conditional subtractions in digit form).  Subtraction adds a multiple of r
// written in redundant digits (every digit >= the largest digit it may have to absorb).  Every function states the bounds it
// needs; the callers (k_pass29 in zk_fr.hip) track them per round.
//
// HBM format is unchanged: canonical 8 x 32-bit Montgomery-form elements with the reference's radix 2^256.  The passes only
// MULTIPLY by a challenge, which the host hands over as c * 2^5 mod r: mont29(c 2^5, x) = c x 2^5 2^-261 = c x 2^-256, the
// product the saturated code computes.  The lazily reduced sums are plain integers (sums of products of representatives):
// any representative gives the same residue, which is what the host reduces them to.
#pragma once
#include "fp.cuh"

namespace zk {

struct F29 {
    u32 l[9];  // value = sum l[i] 2^(29 i); "normalised": l[0..7] < 2^29 (l[8] holds the rest)
};
// 18 digits of a lazily reduced sum of products (each product < 2^522); digits carry up to 3 bits of excess between two
// w29_norm calls (at most 7 accumulations), the top digit the overflow of the per-lane sum (< 2^32: see k_pass29)
struct W29 {
    u32 a[18];
};

struct R29 {
    static constexpr u32 MASK = 0x1fffffffu;
    __host__ __device__ static constexpr u32 P(int i) {  // r
        constexpr u32 t[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
        return t[i];
    }
    // k r in redundant digits: digit 0 borrows 2^29 from digit 1, digits 1..7 borrow from above and lend below, digit 8 lends:
    // every digit below the top is >= 2^29 - 1, the top digit of k r stays above the top digit of any value < (k - 0.8) r
    __host__ __device__ static constexpr u32 K2(int i) {
        constexpr u32 t[9] = {0x20000002u, 0x3fffffefu, 0x3f2dff7eu, 0x36900bfeu, 0x3b00aa76u, 0x380809a0u, 0x2a4199cdu, 0x34ca675eu, 0x00e7db4du};
        return t[i];
    }
    __host__ __device__ static constexpr u32 K3(int i) {
        constexpr u32 t[9] = {0x20000003u, 0x3fffffe7u, 0x3ec4ff3eu, 0x31d811feu, 0x3880ffb2u, 0x240c0e71u, 0x3f6266b5u, 0x3f2f9b0du, 0x015bc8f4u};
        return t[i];
    }
    // 2^261 - k r, normalised: v + (2^261 - k r) reaches 2^261 iff v >= k r
    __host__ __device__ static constexpr u32 C1(int i) {
        constexpr u32 t[9] = {0x1fffffffu, 0x00000007u, 0x00690040u, 0x04b7fa00u, 0x027faac4u, 0x13fbfb2fu, 0x0adf3318u, 0x159acc50u, 0x1f8c1258u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 C2(int i) {
        constexpr u32 t[9] = {0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x1f1824b1u};
        return t[i];
    }
};

// ---- 8 x 32-bit words <-> 9 x 29-bit digits ----
// digit i = bits [29 i, 29 i + 29): one funnel shift + one mask per digit
__device__ __forceinline__ F29 f29_from_words(const u32 (&w)[8]) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, idx = bit >> 5, s = bit & 31;
        u32 v;
        if (s == 0) v = w[idx];
        else if (idx + 1 < 8) v = __builtin_amdgcn_alignbit(w[idx + 1], w[idx], s);
        else v = w[idx] >> s;
        r.l[i] = (i < 8) ? (v & R29::MASK) : v;
    }
    return r;
}
// canonical value (< 2^256, normalised digits) -> words
__device__ __forceinline__ void f29_to_words(const F29& a, u32 (&w)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, d = bit / 29, s = bit - 29 * d;  // word j starts inside digit d at its bit s
        u32 v = a.l[d] >> s;
        if (d + 1 < 9) v |= a.l[d + 1] << (29 - s);
        if (d + 2 < 9 && 58 - s < 32) v |= a.l[d + 2] << (58 - s);
        w[j] = v;
    }
}
__device__ __forceinline__ F29 f29_load(const void* base, size_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + idx * 32);
    const uint4 a = p[0], b = p[1];
    const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return f29_from_words(w);
}

// carry sweep: digits 0..7 back below 2^29 (value unchanged).  Inputs: every digit < 2^32 - 8 so that `+ carry` cannot wrap.
__device__ __forceinline__ void f29_norm(F29& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.l[i + 1] += a.l[i] >> 29;
        a.l[i] &= R29::MASK;
    }
}
// a + k r - b, normalised; a, b normalised, b < (k - 0.8) r.  KR = R29::K2 / K3.
#define ZK_F29_SUB(name, KR)                                                                    \
    __device__ __forceinline__ F29 name(const F29& a, const F29& b) {                          \
        F29 r;                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + R29::KR(i) - b.l[i];   \
        f29_norm(r);                                                                            \
        return r;                                                                               \
    }
ZK_F29_SUB(f29_sub2, K2)
ZK_F29_SUB(f29_sub3, K3)
#undef ZK_F29_SUB
// a + b digit by digit, NOT normalised (digits < 2^30 for normalised inputs): a factor of one wide accumulation
__device__ __forceinline__ F29 f29_add_lazy(const F29& a, const F29& b) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// addend + c x 2^-261 (mod r): the fold f_lo + r (f_hi - f_lo) in one column scan.  c: normalised digits of a value < r
// (wave-uniform: the challenge), x and addend normalised, c x < 2^261 r.  Result normalised, < addend + c x / 2^261 + r.
// Column k holds at most 9 products c_i x_j < 2^58 and 9 products m_i r_j < 2^58: < 2^62.2 with the carry -- no overflow.
// r = 1 mod 2^29, so m_k = -column mod 2^29 and the m_k r_0 term is the addition of m_k.
__device__ __forceinline__ F29 f29_mont_add(const u32 (&c)[9], const F29& x, const F29& addend) {
    u32 m[9];
    F29 t;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) acc += (u64)c[i] * x.l[j];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 9) acc += (u64)m[i] * R29::P(j);
        }
        if (k < 9) {
            const u32 mk = (0u - (u32)acc) & R29::MASK;
            m[k] = mk;
            acc += mk;
        } else {
            acc += addend.l[k - 9];
            t.l[k - 9] = (u32)acc & R29::MASK;
        }
        acc >>= 29;
    }
    t.l[8] = (u32)acc + addend.l[8];
    return t;
}

// w += a b as integers.  Digits of a and b below 2^30 (one lazy addition of normalised values): a column is < 9 x 2^60 +
// carry < 2^63.2.  Every digit of w grows by < 2^29: at most 7 calls between two w29_norm.
__device__ __forceinline__ void w29_mac(W29& w, const F29& a, const F29& b) {
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) acc += (u64)a.l[i] * b.l[j];
        }
        w.a[k] += (u32)acc & R29::MASK;
        acc >>= 29;
    }
    w.a[17] += (u32)acc;
}
__device__ __forceinline__ void w29_norm(W29& w) {
#pragma unroll
    for (int i = 0; i < 17; i++) {
        w.a[i + 1] += w.a[i] >> 29;
        w.a[i] &= R29::MASK;
    }
}

// v (normalised, < 4 r) -> the canonical representative: v - 2r if v >= 2r, then v - r if v >= r
template <int WHICH>
__device__ __forceinline__ void f29_csub(F29& v) {
    F29 t;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 s = v.l[i] + (WHICH == 2 ? R29::C2(i) : R29::C1(i)) + carry;
        carry = s >> 29;
        t.l[i] = s & R29::MASK;
    }
    const bool ge = carry != 0;  // v + 2^261 - k r reached 2^261
#pragma unroll
    for (int i = 0; i < 9; i++) v.l[i] = ge ? t.l[i] : v.l[i];
}
__device__ __forceinline__ void f29_store_canonical(void* base, size_t idx, F29 v) {
    f29_csub<2>(v);
    f29_csub<1>(v);
    u32 w[8];
    f29_to_words(v, w);
    uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + idx * 32);
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

}  // namespace zk
