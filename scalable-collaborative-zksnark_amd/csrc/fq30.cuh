// fq30.cuh -- BLS12-381 Fq for the MSM kernels in UNSATURATED form: 13 limbs of 30 bits in
// registers, Montgomery radix R' = 2^390.
//
// Why: with 32-bit saturated limbs every multiply-accumulate needs a carry instruction
// (v_mad_u64_u32 + v_addc_co_u32: ~7.7 cycles per limb product).  With 30-bit limbs a limb
// product is < 2^60, so up to 15 of them accumulate in a 64-bit register with NO carry handling:
// plain C `acc += (u64)a*b` compiles to back-to-back v_mad_u64_u32 the compiler schedules freely.
// Measured (tools/ubench_mul30.hip, MI355X): 65.7 vs 51.0 G mul/s at 2 waves/SIMD, 60.1 vs 41.7 at 1.
//
// Lazy reduction: q/R' < 2^-9, so a product of inputs < 16q is < 2q -- NO conditional
// subtraction anywhere in the hot path.  Additions / subtractions are limb-wise followed by one
// carry normalisation; subtraction adds a multiple of q written in redundant limbs (every limb
// >= the largest limb it may have to absorb) so no limb ever goes negative.  Callers track the
// static bound of every value (comments "< kq"); everything that is stored stays < 8q < 2^384.
//
// HBM format is unchanged in size and packing (48-byte little-endian integers); only the
// Montgomery constant of INTERNAL data differs (2^390 instead of the reference's 2^384): the SRS is
// converted once at registration (x 2^6) and results are converted back on the way out.
#pragma once
#include "fp.cuh"

namespace zk {

struct Fq30 {
    u32 l[13];  // value = sum l[i] * 2^(30 i); "normalised": l[0..11] < 2^30
};

struct Q30 {
    static constexpr u32 MASK = 0x3fffffffu;
    static constexpr u32 QP = 0x3ffcfffdu;  // -q^{-1} mod 2^30
    __host__ __device__ static constexpr u32 KQ2(int i) {
        constexpr u32 t[13] = {0x7fff5556u, 0x4ff7fffeu, 0x6a7ffff6u, 0x55ffff57u, 0x61ec483cu, 0x469507b4u, 0x6257ece5u, 0x65c279c1u, 0x59aec8edu, 0x7db21a5cu, 0x53496373u, 0x751cbff2u, 0x00340222u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KQ4(int i) {
        constexpr u32 t[13] = {0x7ffeaaacu, 0x5feffffeu, 0x54ffffedu, 0x6bfffeb0u, 0x43d89079u, 0x4d2a0f6au, 0x44afd9cbu, 0x4b84f384u, 0x735d91dcu, 0x7b6434b9u, 0x6692c6e8u, 0x6a397fe5u, 0x00680446u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KQ6(int i) {
        constexpr u32 t[13] = {0x7ffe0002u, 0x6fe7fffeu, 0x7f7fffe4u, 0x41fffe08u, 0x65c4d8b7u, 0x53bf171fu, 0x6707c6b1u, 0x71476d46u, 0x4d0c5acau, 0x79164f17u, 0x79dc2a5du, 0x5f563fd8u, 0x009c066au};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KQ8(int i) {
        constexpr u32 t[13] = {0x7ffd5558u, 0x7fdffffeu, 0x69ffffdbu, 0x57fffd61u, 0x47b120f4u, 0x5a541ed5u, 0x495fb397u, 0x5709e709u, 0x66bb23b9u, 0x76c86974u, 0x4d258dd2u, 0x5472ffccu, 0x00d0088eu};
        return t[i];
    }
    __host__ __device__ static constexpr u32 KQ12(int i) {
        constexpr u32 t[13] = {0x7ffc0004u, 0x5fcffffeu, 0x7effffcau, 0x43fffc12u, 0x4b89b16fu, 0x677e2e40u, 0x4e0f8d63u, 0x628eda8eu, 0x5a18b596u, 0x722c9e2fu, 0x73b854bcu, 0x7eac7fb2u, 0x01380cd5u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 Q(int i) {
        constexpr u32 t[13] = {0x3fffaaabu, 0x27fbffffu, 0x153ffffbu, 0x2affffacu, 0x30f6241eu, 0x034a83dau, 0x112bf673u, 0x12e13ce1u, 0x2cd76477u, 0x1ed90d2eu, 0x29a4b1bau, 0x3a8e5ff9u, 0x001a0111u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 Q2(int i) {
        constexpr u32 t[13] = {0x3fff5556u, 0x0ff7ffffu, 0x2a7ffff7u, 0x15ffff58u, 0x21ec483du, 0x069507b5u, 0x2257ece6u, 0x25c279c2u, 0x19aec8eeu, 0x3db21a5du, 0x13496374u, 0x351cbff3u, 0x00340223u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 Q4(int i) {
        constexpr u32 t[13] = {0x3ffeaaacu, 0x1fefffffu, 0x14ffffeeu, 0x2bfffeb1u, 0x03d8907au, 0x0d2a0f6bu, 0x04afd9ccu, 0x0b84f385u, 0x335d91ddu, 0x3b6434bau, 0x2692c6e9u, 0x2a397fe6u, 0x00680447u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 ONE(int i) {
        constexpr u32 t[13] = {0x00d1ff2eu, 0x19d80000u, 0x34800ac4u, 0x2e00cde6u, 0x02431c84u, 0x269f83a2u, 0x3dcf80ddu, 0x09b42da0u, 0x25eec26cu, 0x15d98f12u, 0x04b29f14u, 0x259fcfa0u, 0x00015de9u};
        return t[i];
    }
    __host__ __device__ static constexpr u32 C396(int i) {
        constexpr u32 t[13] = {0x3480cb7fu, 0x3e0c0000u, 0x2042b126u, 0x3f337aafu, 0x3de4b4d1u, 0x1e015cf1u, 0x005c540du, 0x3467b19au, 0x352a6da3u, 0x19d89d19u, 0x2fb9afe6u, 0x3848c817u, 0x0009772fu};
        return t[i];
    }
    __host__ __device__ static constexpr u32 C384(int i) {
        constexpr u32 t[13] = {0x0002fffdu, 0x18240000u, 0x00c00027u, 0x3d0002f1u, 0x0758baebu, 0x22615d4fu, 0x257455f4u, 0x1614dc14u, 0x2c6d77ceu, 0x2a5e895bu, 0x0935c071u, 0x30fea039u, 0x0015f65eu};
        return t[i];
    }
};

__device__ __forceinline__ Fq30 f30_zero() {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = 0;
    return r;
}
__device__ __forceinline__ Fq30 f30_one() {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = Q30::ONE(i);
    return r;
}
__device__ __forceinline__ bool f30_all_zero(const Fq30& a) {  // exact zero limbs (the infinity marker)
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) o |= a.l[i];
    return o == 0;
}
// carry normalisation: limbs 0..11 back below 2^30 (value unchanged).  Inputs come from one limb-wise
// add/sub of normalised operands: every limb < 3 * 2^30, so `+ carry` (<= 2) cannot wrap a u32.
__device__ __forceinline__ void f30_norm(Fq30& a) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        a.l[i + 1] += a.l[i] >> 30;
        a.l[i] &= Q30::MASK;
    }
}
// a + b   (bounds add)
__device__ __forceinline__ Fq30 f30_add(const Fq30& a, const Fq30& b) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = a.l[i] + b.l[i];
    f30_norm(r);
    return r;
}
// a + 2b  (one pass; limbs < 3 * 2^30 before the carry sweep)
__device__ __forceinline__ Fq30 f30_add2x(const Fq30& a, const Fq30& b) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = a.l[i] + 2u * b.l[i];
    f30_norm(r);
    return r;
}
// a + k*q - b  for a normalised b < k*q.  KQ = redundant limbs of k*q (Q30::KQ2 / KQ4 / KQ8)
#define ZK_F30_SUB(name, KQ)                                                      \
    __device__ __forceinline__ Fq30 name(const Fq30& a, const Fq30& b) {         \
        Fq30 r;                                                                    \
        _Pragma("unroll") for (int i = 0; i < 13; i++) r.l[i] = a.l[i] + Q30::KQ(i) - b.l[i]; \
        f30_norm(r);                                                               \
        return r;                                                                  \
    }
ZK_F30_SUB(f30_sub2, KQ2)  // a + 2q - b,  b < 2q
ZK_F30_SUB(f30_sub4, KQ4)  // a + 4q - b,  b < 4q
ZK_F30_SUB(f30_sub6, KQ6)  // a + 6q - b,  b < 6q
ZK_F30_SUB(f30_sub8, KQ8)  // a + 8q - b,  b < 8q
ZK_F30_SUB(f30_sub12, KQ12)  // a + 12q - b, b < 12q  (the Fq2 formulas of curve30_g2.cuh)
// ---- column budget of the three multipliers -------------------------------------------------------------------
// A column of the product scan holds up to 13 limb products a_i b_j and 13 reduction products m_i q_j: 26 x 2^60 would not
// fit 64 bits, so a column's high part is folded into the next column ("spill": 4 instructions) before it could overflow.
// WHEN is decided at compile time from an upper bound of the column, in units of 2^50:
//     a_i b_j < 2^60 = 1024 units;   (2 a_i) a_j < 2^61 = 2048 units;   m_i q_j < 2^30 q_j = at most (q_j >> 20) + 1 units
// (the limbs of q are constants: together they weigh 6.75 x 2^30, not 13 x 2^30), the carry into a column and the 30 bits kept
// after a spill are below one unit each.  Counting every product as 2^60 cost 10 spills per multiplication; the weighted budget
// needs 4.  f30_*_budget_ok() re-run the same schedule with the LARGEST possible limbs in 128-bit arithmetic and are
// static_assert-ed: no column of any of the three functions can reach 2^64, whatever the (normalised) inputs.
namespace f30b {
static constexpr u32 kLim = 16383, kAB = 1024, kDA = 2048;
__host__ __device__ constexpr u32 mq(int j) { return (Q30::Q(j) >> 20) + 1; }
__host__ __device__ constexpr bool spill(u32 bnd, u32 units) { return bnd + units > kLim; }
}  // namespace f30b
// ---- column order -----------------------------------------------------------------------------------------------------
// A column must START from the carry of the previous one: then every product of the column is one v_mad_u64_u32 whose addend is the
// running sum, and the carry costs nothing.  LLVM's Reassociate pass undoes exactly that: it sorts the leaves of an integer sum by
// "rank" (roughly: how late a value becomes available) and puts the carry -- the end of the previous column's chain -- LAST, so the
// products run as a chain of their own and meet the carry in a separate 64-bit addition (v_lshl_add_u64: one more multiplier-class
// instruction per column, 27 of 448 in f30_mul; 341 per mixed addition in k_accum_tiles).  The rank of a value that comes out of a
// call-like instruction is its POSITION in the block, so the multiplicands of a column are passed through llvm.annotation (returns
// its argument, generates no code, but counts as a call) after ZK_F30_COLUMN() has advanced the position (12 calls: the carry ranks ~6 above the last product of the previous column; 8 is the measured minimum) past the depth of the
// previous chain: the products then rank above the carry and the chain is rebuilt carry-first.  Pure scheduling hint: the value
// computed is the same; tests/test_abi.py checks on the built library that the hint still works with the compiler in use.
#ifndef ZK_F30_NO_CHAIN_HINT
#define ZK_F30_LATE(x) ((u32)__builtin_annotation((unsigned)(x), "zk.f30.column"))
#define ZK_F30_COLUMN()                                                      \
    do {                                                                     \
        _Pragma("unroll") for (int z_ = 0; z_ < 12; z_++) (void)ZK_F30_LATE(0); \
    } while (0)
#else
#define ZK_F30_LATE(x) (x)
#define ZK_F30_COLUMN() do { } while (0)
#endif
#define ZK_F30_ACC(prod, units)                 \
    do {                                        \
        if (f30b::spill(bnd, (units))) {        \
            nxt += acc >> 30;                   \
            acc &= Q30::MASK;                   \
            bnd = 1;                            \
        }                                       \
        acc += (prod);                          \
        bnd += (units);                         \
    } while (0)

// Montgomery product a*b*2^-390 (mod q) for normalised inputs whose bounds multiply to <= 256 q^2;
// the result is normalised and < 2q.  Product scanning, one 64-bit column accumulator (see the budget above).
__device__ __forceinline__ Fq30 f30_mul(const Fq30& a, const Fq30& b) {
    u32 m[13];
    Fq30 t;
    u64 acc = 0;
    u32 bnd = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        u64 nxt = 0;
        ZK_F30_COLUMN();
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (j >= 0 && j < 13) ZK_F30_ACC((u64)ZK_F30_LATE(a.l[i]) * b.l[j], f30b::kAB);
        }
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 13) ZK_F30_ACC((u64)ZK_F30_LATE(m[i]) * Q30::Q(j), f30b::mq(j));
        }
        if (k < 13) {
            const u32 mk = ((u32)acc * Q30::QP) & Q30::MASK;  // (a spill leaves the low 30 bits of the column in place)
            m[k] = mk;
            ZK_F30_ACC((u64)mk * Q30::Q(0), f30b::mq(0));
        } else {
            t.l[k - 13] = (u32)acc & Q30::MASK;
        }
        acc = (acc >> 30) + nxt;
        bnd = 1;
    }
    t.l[12] = (u32)acc;
    return t;
}
// Montgomery square: the 78 off-diagonal products are taken once against the doubled limb (2*a_i < 2^31, product < 2^61).
__device__ __forceinline__ Fq30 f30_sqr(const Fq30& a) {
    u32 m[13], d[13];
    Fq30 t;
#pragma unroll
    for (int i = 0; i < 13; i++) d[i] = a.l[i] << 1;
    u64 acc = 0;
    u32 bnd = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        u64 nxt = 0;
        ZK_F30_COLUMN();
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (j > i && j < 13) ZK_F30_ACC((u64)ZK_F30_LATE(d[i]) * a.l[j], f30b::kDA);
        }
        if ((k & 1) == 0) ZK_F30_ACC((u64)ZK_F30_LATE(a.l[k / 2]) * a.l[k / 2], f30b::kAB);
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 13) ZK_F30_ACC((u64)ZK_F30_LATE(m[i]) * Q30::Q(j), f30b::mq(j));
        }
        if (k < 13) {
            const u32 mk = ((u32)acc * Q30::QP) & Q30::MASK;
            m[k] = mk;
            ZK_F30_ACC((u64)mk * Q30::Q(0), f30b::mq(0));
        } else {
            t.l[k - 13] = (u32)acc & Q30::MASK;
        }
        acc = (acc >> 30) + nxt;
        bnd = 1;
    }
    t.l[12] = (u32)acc;
    return t;
}

// (a*b + c*d) * 2^-390 (mod q) with ONE Montgomery reduction: both products are accumulated column by
// column before the reduction terms (507 mad instead of 676 for two multiplications and an addition).
// Inputs normalised; needs bound(a)*bound(b) + bound(c)*bound(d) <= 256 q^2; result normalised, < 2q.
// A difference of products a*b - c*d is taken as a*b + (kq - c)*d.
__device__ __forceinline__ Fq30 f30_mul2add(const Fq30& a, const Fq30& b, const Fq30& c, const Fq30& d) {
    u32 m[13];
    Fq30 t;
    u64 acc = 0;
    u32 bnd = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        u64 nxt = 0;
        ZK_F30_COLUMN();
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (j >= 0 && j < 13) ZK_F30_ACC((u64)ZK_F30_LATE(a.l[i]) * b.l[j], f30b::kAB);
        }
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (j >= 0 && j < 13) ZK_F30_ACC((u64)ZK_F30_LATE(c.l[i]) * d.l[j], f30b::kAB);
        }
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 13) ZK_F30_ACC((u64)ZK_F30_LATE(m[i]) * Q30::Q(j), f30b::mq(j));
        }
        if (k < 13) {
            const u32 mk = ((u32)acc * Q30::QP) & Q30::MASK;
            m[k] = mk;
            ZK_F30_ACC((u64)mk * Q30::Q(0), f30b::mq(0));
        } else {
            t.l[k - 13] = (u32)acc & Q30::MASK;
        }
        acc = (acc >> 30) + nxt;
        bnd = 1;
    }
    t.l[12] = (u32)acc;
    return t;
}
#undef ZK_F30_ACC
#undef ZK_F30_LATE
#undef ZK_F30_COLUMN

// The same three schedules on WORST-CASE limbs (every variable limb 2^30 - 1, doubled limbs 2^31 - 2, every m_i 2^30 - 1), exact
// 128-bit arithmetic: true iff no column accumulator can reach 2^64.  kind 0: mul, 1: sqr, 2: mul2add.
namespace f30b {
typedef unsigned __int128 u128;
__host__ __device__ constexpr bool budget_ok(int kind) {
    const u128 L = ((u128)1 << 30) - 1, D = 2 * L, LIM64 = (u128)1 << 64;
    u128 acc = 0;
    u32 bnd = 0;
    for (int k = 0; k < 25; k++) {
        u128 nxt = 0;
        // one accumulation step of the schedule: the spill decision of the kernels, the largest product
        auto step = [&](u128 prod, u32 units) {
            if (spill(bnd, units)) {
                nxt += acc >> 30;
                acc = L;  // the 30 bits kept are at most 2^30 - 1
                bnd = 1;
            }
            acc += prod;
            bnd += units;
            return acc < LIM64;
        };
        for (int rep = 0; rep < (kind == 2 ? 2 : 1); rep++)
            for (int i = 0; i < 13; i++) {
                const int j = k - i;
                if (kind == 1) {
                    if (j > i && j < 13 && !step(D * L, kDA)) return false;
                } else if (j >= 0 && j < 13 && !step(L * L, kAB)) return false;
            }
        if (kind == 1 && (k & 1) == 0 && !step(L * L, kAB)) return false;
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 13 && !step(L * Q30::Q(j), mq(j))) return false;
        }
        if (k < 13 && !step(L * Q30::Q(0), mq(0))) return false;
        acc = (acc >> 30) + nxt;
        if (!(acc < ((u128)1 << 50))) return false;  // the carry into a column stays below the one unit it is budgeted with
        bnd = 1;
    }
    return true;
}
static_assert(budget_ok(0) && budget_ok(1) && budget_ok(2), "a column of f30_mul / f30_sqr / f30_mul2add could overflow 64 bits");
}  // namespace f30b

// v - c if v >= c else v, for a normalised constant c given by limb accessor
#define ZK_F30_CSUB(name, C)                                                                         \
    __device__ __forceinline__ Fq30 name(const Fq30& v) {                                           \
        Fq30 d, r;                                                                                    \
        u32 bw = 0;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 13; i++) {                                             \
            u32 x = v.l[i] - Q30::C(i) - bw;                                                          \
            bw = x >> 31;                                                                             \
            d.l[i] = (i < 12) ? (x & Q30::MASK) : x;                                                  \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 13; i++) r.l[i] = bw ? v.l[i] : d.l[i];               \
        return r;                                                                                     \
    }
ZK_F30_CSUB(f30_csub_q, Q)
ZK_F30_CSUB(f30_csub_2q, Q2)
ZK_F30_CSUB(f30_csub_4q, Q4)
// canonical representative (< q) of a normalised value < 8q
__device__ __forceinline__ Fq30 f30_canon8(const Fq30& v) { return f30_csub_q(f30_csub_2q(f30_csub_4q(v))); }
// is a normalised value < 2q congruent to 0?  (only 0 and q qualify)
__device__ __forceinline__ bool f30_is_zero_2q(const Fq30& v) {
    u32 o0 = 0, o1 = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        o0 |= v.l[i];
        o1 |= v.l[i] ^ Q30::Q(i);
    }
    return o0 == 0 || o1 == 0;
}

// ---- 48-byte packed integers in HBM <-> 30-bit limbs (pure re-limbing, value unchanged) ----
__device__ __forceinline__ Fq30 f30_from_words(const u32 (&w)[12]) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int bit = 30 * i, wi = bit >> 5, s = bit & 31;
        u32 v = w[wi] >> s;
        if (s > 2 && wi + 1 < 12) v |= w[wi + 1] << (32 - s);
        r.l[i] = (i < 12) ? (v & Q30::MASK) : v;
    }
    return r;
}
__device__ __forceinline__ void f30_to_words(const Fq30& a, u32 (&w)[12]) {  // a normalised, < 2^384
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int bit = 32 * j, li = bit / 30, s = bit % 30;  // word j starts inside limb li at offset s
        u32 v = a.l[li] >> s;
        if (li + 1 < 13) v |= a.l[li + 1] << (30 - s);
        if (s > 28 && li + 2 < 13) v |= a.l[li + 2] << (60 - s);
        w[j] = v;
    }
}
__device__ __forceinline__ Fq30 f30_load(const void* base, size_t byte_off) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + byte_off);
    u32 w[12];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint4 v = p[i];
        w[4 * i] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
    }
    return f30_from_words(w);
}
__device__ __forceinline__ void f30_store(void* base, size_t byte_off, const Fq30& a) {
    u32 w[12];
    f30_to_words(a, w);
    uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + byte_off);
#pragma unroll
    for (int i = 0; i < 3; i++) p[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// reference form (x * 2^384, canonical) <-> internal form (x * 2^390)
__device__ __forceinline__ Fq30 f30_from_ref(const Fq30& x_r384) {
    Fq30 c;
#pragma unroll
    for (int i = 0; i < 13; i++) c.l[i] = Q30::C396(i);
    return f30_canon8(f30_mul(x_r384, c));  // X * 2^396 * 2^-390 = X * 2^6
}
__device__ __forceinline__ Fq30 f30_to_ref(const Fq30& x_int) {  // input < 16q, output canonical
    Fq30 c;
#pragma unroll
    for (int i = 0; i < 13; i++) c.l[i] = Q30::C384(i);
    return f30_canon8(f30_mul(x_int, c));  // X' * 2^384 * 2^-390 = X' * 2^-6
}
// a^(q-2) (0 -> 0), input < 16q, output < 2q
__device__ __noinline__ Fq30 f30_inv(Fq30 a) {
    Fq30 acc = f30_one();
    for (int i = 383; i >= 0; i--) {
        acc = f30_sqr(acc);
        u32 w = 0;
#pragma unroll
        for (int k = 0; k < 12; k++)
            if ((i >> 5) == k) w = fp_pm2_limb<FqCfg>(k);
        if ((w >> (i & 31)) & 1u) acc = f30_mul(acc, a);
    }
    return acc;
}

}  // namespace zk
