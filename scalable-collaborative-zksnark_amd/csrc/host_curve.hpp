// host_curve.hpp -- host-side (CPU) Fq / G1 arithmetic used by libzkhip for the O(255) serial
// tail of an MSM (combining <= ~300 per-bit bucket sums by double-and-add and normalising to
// affine) and for SRS packing / synthetic-SRS generation.  It is product code (not the oracle):
// the serial tail is a dependency chain of ~255 doublings that would cost >1.5 ms on one GPU
// lane and ~0.1 ms here, so it is done on the host like every GPU MSM library does.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

namespace zkhost {

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef std::array<u64, 6> Fq;

static const Fq Q = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                     0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 QINV = 0x89f3fffcfffcfffdULL;
static const Fq ONE = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                       0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const Fq R2 = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                      0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
static const Fq ZERO = {0, 0, 0, 0, 0, 0};
// 2^378 (< q): the 2^384-Montgomery form of 2^-6, converts device-internal values (radix 2^390) back
static inline const Fq& K378() {
    static const Fq k = {0, 0, 0, 0, 0, 0x0400000000000000ULL};
    return k;
}
// generator of G1 (affine, canonical): ark-bls12-381 G1_GENERATOR_X / _Y
static const Fq GX_CANON = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                            0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const Fq GY_CANON = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                            0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};

static inline bool is_zero(const Fq& a) {
    u64 o = 0;
    for (int i = 0; i < 6; i++) o |= a[i];
    return o == 0;
}
static inline bool geq(const Fq& a, const Fq& b) {
    for (int i = 5; i >= 0; i--) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
static inline Fq add(const Fq& a, const Fq& b) {
    Fq r;
    u64 c = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a[i] + b[i] + c;
        r[i] = (u64)t;
        c = (u64)(t >> 64);
    }
    if (c || geq(r, Q)) {
        u64 bw = 0;
        for (int i = 0; i < 6; i++) {
            u128 t = (u128)r[i] - Q[i] - bw;
            r[i] = (u64)t;
            bw = (u64)(t >> 64) & 1;
        }
    }
    return r;
}
static inline Fq sub(const Fq& a, const Fq& b) {
    Fq r;
    u64 bw = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a[i] - b[i] - bw;
        r[i] = (u64)t;
        bw = (u64)(t >> 64) & 1;
    }
    if (bw) {
        u64 c = 0;
        for (int i = 0; i < 6; i++) {
            u128 t = (u128)r[i] + Q[i] + c;
            r[i] = (u64)t;
            c = (u64)(t >> 64);
        }
    }
    return r;
}
static inline Fq neg(const Fq& a) { return is_zero(a) ? a : sub(ZERO, a); }
static inline Fq dbl(const Fq& a) { return add(a, a); }

#define ZKH_MM(k) \
    x = (u128)a[k] * bi + t##k + c; \
    t##k = (u64)x; \
    c = (u64)(x >> 64);
#define ZKH_MR(k, km) \
    x = (u128)m * Q[k] + t##k + c; \
    t##km = (u64)x; \
    c = (u64)(x >> 64);
static inline Fq mul(const Fq& a, const Fq& b) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#pragma GCC unroll 6
    for (int i = 0; i < 6; i++) {
        u64 bi = b[i], c = 0;
        u128 x;
        ZKH_MM(0) ZKH_MM(1) ZKH_MM(2) ZKH_MM(3) ZKH_MM(4) ZKH_MM(5)
        t6 += c;  // q < 2^381: the running value stays below 2q, no carry out of t6
        u64 m = t0 * QINV;
        x = (u128)m * Q[0] + t0;
        c = (u64)(x >> 64);
        ZKH_MR(1, 0) ZKH_MR(2, 1) ZKH_MR(3, 2) ZKH_MR(4, 3) ZKH_MR(5, 4)
        x = (u128)t6 + c;
        t5 = (u64)x;
        t6 = (u64)(x >> 64);
    }
    Fq r = {t0, t1, t2, t3, t4, t5};
    if (t6 || geq(r, Q)) {
        u64 bw = 0;
        for (int i = 0; i < 6; i++) {
            u128 t = (u128)r[i] - Q[i] - bw;
            r[i] = (u64)t;
            bw = (u64)(t >> 64) & 1;
        }
    }
    return r;
}
#undef ZKH_MM
#undef ZKH_MR
static inline Fq sqr(const Fq& a) { return mul(a, a); }
static inline Fq to_mont(const Fq& a) { return mul(a, R2); }
static inline Fq inv(const Fq& a) {  // a^(q-2)
    Fq e = Q;
    e[0] -= 2;
    Fq acc = ONE;
    for (int i = 383; i >= 0; i--) {
        acc = sqr(acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = mul(acc, a);
    }
    return acc;
}

// ---- Fq2 = Fq[u] / (u^2 + 1): the coordinate field of G2 ----
struct Fq2 {
    Fq c0, c1;
};
static inline bool operator==(const Fq2& a, const Fq2& b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline bool is_zero(const Fq2& a) { return is_zero(a.c0) && is_zero(a.c1); }
static inline Fq2 add(const Fq2& a, const Fq2& b) { return Fq2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
static inline Fq2 sub(const Fq2& a, const Fq2& b) { return Fq2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
static inline Fq2 neg(const Fq2& a) { return Fq2{neg(a.c0), neg(a.c1)}; }
static inline Fq2 dbl(const Fq2& a) { return add(a, a); }
static inline Fq2 mul(const Fq2& a, const Fq2& b) {  // Karatsuba: 3 multiplications
    const Fq t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    const Fq t2 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fq2{sub(t0, t1), sub(sub(t2, t0), t1)};
}
static inline Fq2 sqr(const Fq2& a) { return Fq2{mul(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(mul(a.c0, a.c1))}; }
static inline Fq2 inv(const Fq2& a) {  // conj(a) / (a0^2 + a1^2)
    const Fq n = inv(add(sqr(a.c0), sqr(a.c1)));
    return Fq2{mul(a.c0, n), neg(mul(a.c1, n))};
}
template <class F>
struct FieldConst;
template <>
struct FieldConst<Fq> {
    static Fq one() { return ONE; }
    static Fq zero() { return ZERO; }
};
template <>
struct FieldConst<Fq2> {
    static Fq2 one() { return Fq2{ONE, ZERO}; }
    static Fq2 zero() { return Fq2{ZERO, ZERO}; }
};

// ---- short Weierstrass curves with a = 0 over F (G1: F = Fq, G2: F = Fq2) ----
template <class F>
struct AffT {
    F x, y;  // x = y = 0: infinity
};
template <class F>
struct JacT {
    F x, y, z;  // z = 0: infinity
};
typedef AffT<Fq> Aff;
typedef JacT<Fq> Jac;
typedef AffT<Fq2> Aff2;
typedef JacT<Fq2> Jac2;
template <class F>
static inline JacT<F> jac_inf_t() {
    return JacT<F>{FieldConst<F>::one(), FieldConst<F>::one(), FieldConst<F>::zero()};
}
static inline Jac jac_inf() { return jac_inf_t<Fq>(); }
template <class F>
static inline bool aff_inf(const AffT<F>& p) {
    return is_zero(p.x) && is_zero(p.y);
}

template <class F>
static inline JacT<F> jac_dbl(const JacT<F>& p) {  // dbl-2009-l
    if (is_zero(p.z)) return p;
    F A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    F t = add(p.x, B);
    F D = dbl(sub(sub(sqr(t), A), C));
    F E = add(dbl(A), A);
    F Fv = sqr(E);
    JacT<F> r;
    r.x = sub(Fv, dbl(D));
    r.z = dbl(mul(p.y, p.z));
    F C8 = dbl(dbl(dbl(C)));
    r.y = sub(mul(E, sub(D, r.x)), C8);
    return r;
}
template <class F>
static inline JacT<F> jac_add(const JacT<F>& p, const JacT<F>& q) {  // add-2007-bl
    if (is_zero(p.z)) return q;
    if (is_zero(q.z)) return p;
    F Z1Z1 = sqr(p.z), Z2Z2 = sqr(q.z);
    F U1 = mul(p.x, Z2Z2), U2 = mul(q.x, Z1Z1);
    F S1 = mul(mul(p.y, q.z), Z2Z2), S2 = mul(mul(q.y, p.z), Z1Z1);
    if (U1 == U2) {
        if (S1 == S2) return jac_dbl(p);
        return jac_inf_t<F>();
    }
    F H = sub(U2, U1);
    F I = sqr(dbl(H));
    F J = mul(H, I);
    F rr = dbl(sub(S2, S1));
    F V = mul(U1, I);
    JacT<F> r;
    r.x = sub(sub(sqr(rr), J), dbl(V));
    r.y = sub(mul(rr, sub(V, r.x)), dbl(mul(S1, J)));
    r.z = mul(sub(sub(sqr(add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}
template <class F>
static inline JacT<F> jac_add_mixed(const JacT<F>& p, const AffT<F>& q) {  // madd-2007-bl
    if (aff_inf(q)) return p;
    if (is_zero(p.z)) return JacT<F>{q.x, q.y, FieldConst<F>::one()};
    F Z1Z1 = sqr(p.z);
    F U2 = mul(q.x, Z1Z1);
    F S2 = mul(mul(q.y, p.z), Z1Z1);
    if (U2 == p.x) {
        if (S2 == p.y) return jac_dbl(p);
        return jac_inf_t<F>();
    }
    F H = sub(U2, p.x);
    F HH = sqr(H);
    F I = dbl(dbl(HH));
    F J = mul(H, I);
    F rr = dbl(sub(S2, p.y));
    F V = mul(p.x, I);
    JacT<F> r;
    r.x = sub(sub(sqr(rr), J), dbl(V));
    r.y = sub(mul(rr, sub(V, r.x)), dbl(mul(p.y, J)));
    r.z = sub(sub(sqr(add(p.z, H)), Z1Z1), HH);
    return r;
}
// XYZZ (x = X/ZZ, y = Y/ZZZ) -> the Jacobian representative (X*ZZ, Y*ZZZ, ZZ): x = X'/Z'^2, y = Y'/Z'^3
// (ZZZ^2 = ZZ^3); no inversion needed.
static inline Jac xyzz_to_jac(const Fq& X, const Fq& Y, const Fq& ZZ, const Fq& ZZZ) {
    if (is_zero(ZZ)) return jac_inf();
    return Jac{mul(X, ZZ), mul(Y, ZZZ), ZZ};
}
template <class F>
static inline AffT<F> jac_to_aff(const JacT<F>& p) {
    if (is_zero(p.z)) return AffT<F>{FieldConst<F>::zero(), FieldConst<F>::zero()};
    F zi = inv(p.z);
    F zi2 = sqr(zi);
    return AffT<F>{mul(p.x, zi2), mul(p.y, mul(zi2, zi))};
}
static inline void put_fe(const Fq& v, uint64_t* out) { memcpy(out, v.data(), 48); }
static inline void put_fe(const Fq2& v, uint64_t* out) {
    memcpy(out, v.c0.data(), 48);
    memcpy(out + 6, v.c1.data(), 48);
}
static inline void get_fe(Fq& v, const uint64_t* in) { memcpy(v.data(), in, 48); }
static inline void get_fe(Fq2& v, const uint64_t* in) {
    memcpy(v.c0.data(), in, 48);
    memcpy(v.c1.data(), in + 6, 48);
}
template <class F>
constexpr size_t fe_words() {
    return sizeof(F) / 8;
}
// normalised Jacobian in ark-ec's Projective layout {x, y, z}: (x, y, 1) or (1, 1, 0) for infinity; 18 u64 (G1) / 36 (G2)
template <class F>
static inline void write_normalised(const JacT<F>& p, uint64_t* out) {
    constexpr size_t w = fe_words<F>();
    if (is_zero(p.z)) {
        put_fe(FieldConst<F>::one(), out);
        put_fe(FieldConst<F>::one(), out + w);
        put_fe(FieldConst<F>::zero(), out + 2 * w);
        return;
    }
    AffT<F> a = jac_to_aff(p);
    put_fe(a.x, out);
    put_fe(a.y, out + w);
    put_fe(FieldConst<F>::one(), out + 2 * w);
}
// k*P, k = 4 canonical u64 limbs
template <class F>
static inline JacT<F> scalar_mul(const AffT<F>& P, const uint64_t k[4]) {
    JacT<F> acc = jac_inf_t<F>();
    for (int i = 255; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((k[i / 64] >> (i % 64)) & 1) acc = jac_add_mixed(acc, P);
    }
    return acc;
}
// k*P for a Jacobian P
template <class F>
static inline JacT<F> scalar_mul_jac(const JacT<F>& P, const uint64_t k[4]) {
    JacT<F> acc = jac_inf_t<F>();
    for (int i = 255; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((k[i / 64] >> (i % 64)) & 1) acc = jac_add(acc, P);
    }
    return acc;
}
// batch normalisation (one inversion)
template <class F>
static inline void batch_to_affine(const std::vector<JacT<F>>& in, AffT<F>* out) {
    size_t n = in.size();
    std::vector<F> pref(n);
    F acc = FieldConst<F>::one();
    for (size_t i = 0; i < n; i++) {
        pref[i] = acc;
        if (!is_zero(in[i].z)) acc = mul(acc, in[i].z);
    }
    F iv = inv(acc);
    for (size_t i = n; i-- > 0;) {
        if (is_zero(in[i].z)) {
            out[i] = AffT<F>{FieldConst<F>::zero(), FieldConst<F>::zero()};
            continue;
        }
        F zi = mul(iv, pref[i]);
        iv = mul(iv, in[i].z);
        F zi2 = sqr(zi);
        out[i] = AffT<F>{mul(in[i].x, zi2), mul(in[i].y, mul(zi2, zi))};
    }
}

}  // namespace zkhost
