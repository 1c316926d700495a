// curve30_g2.cuh -- BLS12-381 G2 (the twist over Fq2 = Fq[u] / (u^2 + 1)) on the unsaturated field of fq30.cuh:
// the same XYZZ formulas and the same static bounds as curve30.cuh, every coordinate a pair of Fq30.
//
// `d_msm` is generic over CurveGroup (dist-primitive/src/dmsm.rs:9); the reference's parameters carry G2 points
// in powers_of_g2 (dpoly_comm.rs:27,59-62).  This header gives the MSM pipeline of zk_msm.hip its second curve.
//
// Fq2 multiplication = two fused two-product Montgomery multiplications (f30_mul2add): c0 = a0 b0 + a1 (k q - b1),
// c1 = a0 b1 + a1 b0 -- two reductions per product instead of the three of Karatsuba-with-reductions.  Bound rule:
// bound(a) * (bound(b) + k) <= 256 with k >= bound(b) the redundant multiple of q used for the negation.
// Invariants per component (as in curve30.cuh): affine x, y < q; XYZZ X < 8q, Y < 4q, ZZ < 2q, ZZZ < 2q.
#pragma once
#include "fq30.cuh"

namespace zk {

struct Fq2x {
    Fq30 c0, c1;
};
struct Aff2 {
    Fq2x x, y;
};
struct Xyzz2 {
    Fq2x x, y, zz, zzz;
};

__device__ __forceinline__ Fq2x f2_zero() { return Fq2x{f30_zero(), f30_zero()}; }
__device__ __forceinline__ Fq2x f2_one() { return Fq2x{f30_one(), f30_zero()}; }
__device__ __forceinline__ bool f2_all_zero(const Fq2x& a) { return f30_all_zero(a.c0) && f30_all_zero(a.c1); }
__device__ __forceinline__ bool f2_is_zero_2q(const Fq2x& a) { return f30_is_zero_2q(a.c0) && f30_is_zero_2q(a.c1); }
__device__ __forceinline__ Fq2x f2_add(const Fq2x& a, const Fq2x& b) { return Fq2x{f30_add(a.c0, b.c0), f30_add(a.c1, b.c1)}; }
__device__ __forceinline__ Fq2x f2_add2x(const Fq2x& a, const Fq2x& b) { return Fq2x{f30_add2x(a.c0, b.c0), f30_add2x(a.c1, b.c1)}; }
#define ZK_F2_SUB(name, sub) \
    __device__ __forceinline__ Fq2x name(const Fq2x& a, const Fq2x& b) { return Fq2x{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
ZK_F2_SUB(f2_sub2, f30_sub2)
ZK_F2_SUB(f2_sub4, f30_sub4)
ZK_F2_SUB(f2_sub6, f30_sub6)
ZK_F2_SUB(f2_sub8, f30_sub8)
// k q - x for x < k q
template <int K>
__device__ __forceinline__ Fq30 f30_negk(const Fq30& x) {
    static_assert(K == 2 || K == 4 || K == 6 || K == 8 || K == 12, "available redundant multiples of q");
    return K == 2 ? f30_sub2(f30_zero(), x) : K == 4 ? f30_sub4(f30_zero(), x) : K == 6 ? f30_sub6(f30_zero(), x) : K == 8 ? f30_sub8(f30_zero(), x) : f30_sub12(f30_zero(), x);
}
// a * b, b < KB q: needs bound(a) * (bound(b) + KB) <= 256; result < 2q per component
template <int KB>
__device__ __forceinline__ Fq2x f2_mul(const Fq2x& a, const Fq2x& b) {
    Fq2x r;
    r.c0 = f30_mul2add(a.c0, b.c0, a.c1, f30_negk<KB>(b.c1));  // a0 b0 - a1 b1
    r.c1 = f30_mul2add(a.c0, b.c1, a.c1, b.c0);
    return r;
}
// a^2, a < KA q: bound(a) * (bound(a) + KA) <= 256
template <int KA>
__device__ __forceinline__ Fq2x f2_sqr(const Fq2x& a) {
    Fq2x r;
    r.c0 = f30_mul2add(a.c0, a.c0, a.c1, f30_negk<KA>(a.c1));  // a0^2 - a1^2
    r.c1 = f30_mul2add(a.c0, a.c1, a.c0, a.c1);                // 2 a0 a1
    return r;
}
__device__ __forceinline__ Fq2x f2_canon8(const Fq2x& a) { return Fq2x{f30_canon8(a.c0), f30_canon8(a.c1)}; }
__device__ __forceinline__ Fq2x f2_neg_canon(const Fq2x& y) { return Fq2x{f30_neg_canon(y.c0), f30_neg_canon(y.c1)}; }

__device__ __forceinline__ bool aff2_is_inf(const Aff2& p) { return f2_all_zero(p.x) && f2_all_zero(p.y); }
__device__ __forceinline__ bool xyzz2_is_inf(const Xyzz2& p) { return f2_all_zero(p.zz); }
__device__ __forceinline__ void xyzz2_set_inf(Xyzz2& p) { p.x = p.y = p.zz = p.zzz = f2_zero(); }

// affine records: 192 bytes, x.c0 | x.c1 | y.c0 | y.c1 (48-byte integers, internal Montgomery form)
__device__ __forceinline__ Aff2 aff2_load(const void* base, size_t idx) {
    Aff2 p;
    p.x.c0 = f30_load(base, idx * 192);
    p.x.c1 = f30_load(base, idx * 192 + 48);
    p.y.c0 = f30_load(base, idx * 192 + 96);
    p.y.c1 = f30_load(base, idx * 192 + 144);
    return p;
}
// XYZZ arrays: blocks of 64 points, 24 chunks of 16 bytes per point side by side (curve30.cuh, twice the chunks)
__device__ __forceinline__ size_t xyzz2_chunk_off(size_t idx, int k) { return (idx >> 6) * 24576 + (size_t)k * 1024 + (idx & 63) * 16; }
__device__ __forceinline__ Fq30 f30_load_chunks2(const void* base, size_t idx, int k0) {
    const char* b = reinterpret_cast<const char*>(base);
    u32 w[12];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint4 v = *reinterpret_cast<const uint4*>(b + xyzz2_chunk_off(idx, k0 + i));
        w[4 * i] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
    }
    return f30_from_words(w);
}
__device__ __forceinline__ void f30_store_chunks2(void* base, size_t idx, int k0, const Fq30& a) {
    char* b = reinterpret_cast<char*>(base);
    u32 w[12];
    f30_to_words(a, w);
#pragma unroll
    for (int i = 0; i < 3; i++)
        *reinterpret_cast<uint4*>(b + xyzz2_chunk_off(idx, k0 + i)) = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ Fq2x f2_load_chunks(const void* base, size_t idx, int k0) {
    return Fq2x{f30_load_chunks2(base, idx, k0), f30_load_chunks2(base, idx, k0 + 3)};
}
__device__ __forceinline__ void f2_store_chunks(void* base, size_t idx, int k0, const Fq2x& a) {
    f30_store_chunks2(base, idx, k0, a.c0);
    f30_store_chunks2(base, idx, k0 + 3, a.c1);
}
__device__ __forceinline__ Xyzz2 xyzz2_load(const void* base, size_t idx) {
    Xyzz2 p;
    p.x = f2_load_chunks(base, idx, 0);
    p.y = f2_load_chunks(base, idx, 6);
    p.zz = f2_load_chunks(base, idx, 12);
    p.zzz = f2_load_chunks(base, idx, 18);
    return p;
}
__device__ __forceinline__ void xyzz2_store(void* base, size_t idx, const Xyzz2& p) {
    f2_store_chunks(base, idx, 0, p.x);
    f2_store_chunks(base, idx, 6, p.y);
    f2_store_chunks(base, idx, 12, p.zz);
    f2_store_chunks(base, idx, 18, p.zzz);
}

// 2 * p (dbl-2008-s-1, a = 0)
__device__ __forceinline__ Xyzz2 xyzz2_dbl(const Xyzz2& p) {
    if (xyzz2_is_inf(p)) return p;
    Xyzz2 r;
    const Fq2x U = f2_add(p.y, p.y);                    // < 8q
    const Fq2x V = f2_sqr<8>(U);                        // 8 * 16
    const Fq2x W = f2_mul<2>(U, V);                     // 8 * 4
    const Fq2x S = f2_mul<2>(p.x, V);                   // 8 * 4
    const Fq2x xx = f2_sqr<8>(p.x);                     // 8 * 16
    const Fq2x M = f2_add(f2_add(xx, xx), xx);          // < 6q
    const Fq2x S2 = f2_add(S, S);                       // < 4q
    const Fq2x X3 = f2_sub4(f2_sqr<6>(M), S2);          // M^2 + 4q - 2S < 6q
    r.y = f2_sub2(f2_mul<12>(M, f2_sub8(S, X3)), f2_mul<4>(W, p.y));  // 6 * 22, 2 * 8: < 4q
    r.x = X3;
    r.zz = f2_mul<2>(V, p.zz);
    r.zzz = f2_mul<2>(W, p.zzz);
    return r;
}
__device__ __forceinline__ Xyzz2 xyzz2_dbl_affine(const Fq2x& x, const Fq2x& y) {
    Xyzz2 p;
    p.x = x;
    p.y = y;
    p.zz = f2_one();
    p.zzz = f2_one();
    return xyzz2_dbl(p);
}

// acc += p (affine; `neg`: -p), madd-2008-s with the bounds of curve30.cuh
__device__ __forceinline__ void xyzz2_madd(Xyzz2& acc, const Aff2& p, bool neg) {
    if (aff2_is_inf(p)) return;
    if (xyzz2_is_inf(acc)) {
        acc.x = p.x;
        acc.y = neg ? f2_neg_canon(p.y) : p.y;
        acc.zz = f2_one();
        acc.zzz = f2_one();
        return;
    }
    const Fq2x U2 = f2_mul<2>(p.x, acc.zz);    // 1 * 4
    const Fq2x S2 = f2_mul<2>(p.y, acc.zzz);
    const Fq2x P = f2_sub8(U2, acc.x);         // < 10q
    const Fq2x R = neg ? f2_sub2(f2_sub4(f2_zero(), acc.y), S2) : f2_sub4(S2, acc.y);  // < 6q
    const Fq2x PP = f2_sqr<12>(P);             // 10 * 22 = 220
    const Fq2x PPP = f2_mul<2>(P, PP);         // 10 * 4
    const Fq2x Q = f2_mul<2>(acc.x, PP);       // 8 * 4
    const Fq2x ZZ3 = f2_mul<2>(acc.zz, PP);
    if (f2_is_zero_2q(ZZ3)) {                  // P = 0: same x, doubling or cancellation
        const Fq2x Rc = f2_canon8(R);
        if (f2_all_zero(Rc)) acc = xyzz2_dbl_affine(p.x, neg ? f2_neg_canon(p.y) : p.y);
        else xyzz2_set_inf(acc);
        return;
    }
    const Fq2x X3 = f2_sub6(f2_sqr<6>(R), f2_add2x(PPP, Q));          // R^2 + 6q - (PPP + 2Q) < 8q
    const Fq2x T = f2_mul<12>(R, f2_sub8(Q, X3));                       // 6 * 22
    acc.y = f2_sub2(T, f2_mul<2>(acc.y, PPP));                          // < 4q
    acc.zzz = f2_mul<2>(acc.zzz, PPP);
    acc.zz = ZZ3;
    acc.x = X3;
}

// a + b (add-2008-s)
__device__ __forceinline__ Xyzz2 xyzz2_add(const Xyzz2& a, const Xyzz2& b) {
    if (xyzz2_is_inf(a)) return b;
    if (xyzz2_is_inf(b)) return a;
    const Fq2x U1 = f2_mul<2>(a.x, b.zz);      // 8 * 4
    const Fq2x U2 = f2_mul<2>(b.x, a.zz);
    const Fq2x S1 = f2_mul<2>(a.y, b.zzz);     // 4 * 4
    const Fq2x S2 = f2_mul<2>(b.y, a.zzz);
    const Fq2x P = f2_sub2(U2, U1);            // < 4q
    const Fq2x R = f2_sub2(S2, S1);            // < 4q
    const Fq2x PP = f2_sqr<4>(P);              // 4 * 8
    const Fq2x PPP = f2_mul<2>(P, PP);
    const Fq2x Q = f2_mul<2>(U1, PP);
    const Fq2x ZZ3 = f2_mul<2>(f2_mul<2>(a.zz, b.zz), PP);
    Xyzz2 r;
    if (f2_is_zero_2q(ZZ3)) {
        if (f2_all_zero(f2_canon8(R))) r = xyzz2_dbl(a);
        else xyzz2_set_inf(r);
        return r;
    }
    r.x = f2_sub6(f2_sqr<4>(R), f2_add2x(PPP, Q));                     // < 8q
    r.y = f2_sub2(f2_mul<12>(R, f2_sub8(Q, r.x)), f2_mul<2>(S1, PPP));  // 4 * 22; < 4q
    r.zz = ZZ3;
    r.zzz = f2_mul<2>(f2_mul<2>(a.zzz, b.zzz), PPP);
    return r;
}

// ---- one G2 XYZZ addition spread over the four lanes of a quad: the rounds of curve30.cuh's xyzz30_add_quad, every
// operand a pair of Fq30 (an Fq2 multiplication per lane and round).  The late reduction passes and the fix-up of a G2 MSM
// are chains of single additions: ~3 x the latency of G1's per multiplication, so the quad form matters more here. ----
template <int R>
__device__ __forceinline__ Fq2x f2_quad_bcast(const Fq2x& v) {
    Fq2x r;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        r.c0.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)v.c0.l[i], R * 0x55, 0xf, 0xf, true);
        asm volatile("" : "+v"(r.c0.l[i]));  // (keep it a plain v_mov_b32_dpp: see curve30.cuh)
        r.c1.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)v.c1.l[i], R * 0x55, 0xf, 0xf, true);
        asm volatile("" : "+v"(r.c1.l[i]));
    }
    return r;
}
__device__ __forceinline__ Fq2x f2_sel4(int role, const Fq2x& a0, const Fq2x& a1, const Fq2x& a2, const Fq2x& a3) {
    Fq2x r;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        r.c0.l[i] = (role == 0) ? a0.c0.l[i] : (role == 1) ? a1.c0.l[i] : (role == 2) ? a2.c0.l[i] : a3.c0.l[i];
        r.c1.l[i] = (role == 0) ? a0.c1.l[i] : (role == 1) ? a1.c1.l[i] : (role == 2) ? a2.c1.l[i] : a3.c1.l[i];
    }
    return r;
}
// out[io] = in[ia] + in[ib]; all four lanes of a quad call it with identical indices (role = lane & 3).
// Round 1: U1 = X1*ZZ2 | U2 = X2*ZZ1 | S1 = Y1*ZZZ2 | S2 = Y2*ZZZ1      bounds 8q * (2q + 2q)
// Round 2: A = ZZ1*ZZ2 | B = ZZZ1*ZZZ2 | PP = P^2 | RR = R^2             4q * (4q + 4q)
// Round 3: Q = U1*PP   | ZZ3 = A*PP    | PPP = P*PP | -                  4q * (2q + 2q)
// Round 4: -           | ZZZ3 = B*PPP  | T1 = S1*PPP | T2 = R*(Q + 8q - X3)   4q * (10q + 12q);  Y3 = T2 - T1
__device__ __forceinline__ void xyzz2_add_quad(const void* __restrict__ in, size_t ia, size_t ib, void* __restrict__ out, size_t io, int role) {
    const Fq2x opA = f2_load_chunks(in, (role & 1) ? ib : ia, (role & 2) ? 6 : 0);    // a.x | b.x | a.y | b.y
    const Fq2x opB = f2_load_chunks(in, (role & 1) ? ia : ib, (role & 2) ? 18 : 12);  // b.zz | a.zz | b.zzz | a.zzz
    const Fq2x zz2 = f2_quad_bcast<0>(opB), zz1 = f2_quad_bcast<1>(opB), zzz2 = f2_quad_bcast<2>(opB), zzz1 = f2_quad_bcast<3>(opB);
    const bool a_inf = f2_all_zero(zz1), b_inf = f2_all_zero(zz2);  // identical in the four lanes
    if (a_inf || b_inf) {  // the other operand (or infinity) is the result: lane r copies coordinate r
        f2_store_chunks(out, io, 6 * role, f2_load_chunks(in, a_inf ? ib : ia, 6 * role));
        return;
    }
    const Fq2x m1 = f2_mul<2>(opA, opB);
    const Fq2x u1 = f2_quad_bcast<0>(m1), u2 = f2_quad_bcast<1>(m1), s1 = f2_quad_bcast<2>(m1), s2 = f2_quad_bcast<3>(m1);
    const Fq2x P = f2_sub2(u2, u1);  // < 4q
    const Fq2x R = f2_sub2(s2, s1);  // < 4q
    const Fq2x m2 = f2_mul<4>(f2_sel4(role, zz1, zzz1, P, R), f2_sel4(role, zz2, zzz2, P, R));
    const Fq2x A = f2_quad_bcast<0>(m2), B = f2_quad_bcast<1>(m2), PP = f2_quad_bcast<2>(m2), RR = f2_quad_bcast<3>(m2);
    const Fq2x m3 = f2_mul<2>(f2_sel4(role, u1, A, P, P), PP);
    const Fq2x Q = f2_quad_bcast<0>(m3), ZZ3 = f2_quad_bcast<1>(m3), PPP = f2_quad_bcast<2>(m3);
    if (f2_is_zero_2q(ZZ3)) {  // same x: doubling or cancellation (adversarial inputs only) -- lane 0 redoes it alone
        if (role == 0) xyzz2_store(out, io, xyzz2_add(xyzz2_load(in, ia), xyzz2_load(in, ib)));
        return;
    }
    const Fq2x X3 = f2_sub6(RR, f2_add2x(PPP, Q));  // < 8q
    const Fq2x m4 = f2_mul<12>(f2_sel4(role, B, B, s1, R), f2_sel4(role, PPP, PPP, PPP, f2_sub8(Q, X3)));
    const Fq2x ZZZ3 = f2_quad_bcast<1>(m4), T1 = f2_quad_bcast<2>(m4), T2 = f2_quad_bcast<3>(m4);
    const Fq2x Y3 = f2_sub2(T2, T1);  // < 4q
    f2_store_chunks(out, io, 6 * role, f2_sel4(role, X3, Y3, ZZ3, ZZZ3));
}
// acc += in[ib], acc replicated in the registers of the four lanes (the fix-up's chain), same rounds
__device__ __forceinline__ Xyzz2 xyzz2_acc_quad(const Xyzz2& acc, const void* __restrict__ in, size_t ib, int role) {
    const Fq2x bpart = f2_load_chunks(in, ib, role == 0 ? 12 : role == 1 ? 0 : role == 2 ? 18 : 6);  // b.zz | b.x | b.zzz | b.y
    const Fq2x zz2 = f2_quad_bcast<0>(bpart);
    if (f2_all_zero(zz2)) return acc;                    // b is infinity
    if (xyzz2_is_inf(acc)) return xyzz2_load(in, ib);    // (all lanes load all of b)
    const Fq2x zzz2 = f2_quad_bcast<2>(bpart);
    // U1 = X1*ZZ2 | U2 = X2*ZZ1 | S1 = Y1*ZZZ2 | S2 = Y2*ZZZ1: first operand < 8q, second < 2q
    const Fq2x m1 = f2_mul<2>(f2_sel4(role, acc.x, bpart, acc.y, bpart), f2_sel4(role, bpart, acc.zz, bpart, acc.zzz));
    const Fq2x u1 = f2_quad_bcast<0>(m1), u2 = f2_quad_bcast<1>(m1), s1 = f2_quad_bcast<2>(m1), s2 = f2_quad_bcast<3>(m1);
    const Fq2x P = f2_sub2(u2, u1);
    const Fq2x R = f2_sub2(s2, s1);
    const Fq2x m2 = f2_mul<4>(f2_sel4(role, acc.zz, acc.zzz, P, R), f2_sel4(role, zz2, zzz2, P, R));
    const Fq2x A = f2_quad_bcast<0>(m2), B = f2_quad_bcast<1>(m2), PP = f2_quad_bcast<2>(m2), RR = f2_quad_bcast<3>(m2);
    const Fq2x m3 = f2_mul<2>(f2_sel4(role, u1, A, P, P), PP);
    const Fq2x Q = f2_quad_bcast<0>(m3), ZZ3 = f2_quad_bcast<1>(m3), PPP = f2_quad_bcast<2>(m3);
    if (f2_is_zero_2q(ZZ3)) return xyzz2_add(acc, xyzz2_load(in, ib));  // doubling / cancellation: every lane alone, same result
    Xyzz2 r;
    r.x = f2_sub6(RR, f2_add2x(PPP, Q));
    const Fq2x m4 = f2_mul<12>(f2_sel4(role, B, B, s1, R), f2_sel4(role, PPP, PPP, PPP, f2_sub8(Q, r.x)));
    r.zzz = f2_quad_bcast<1>(m4);
    r.y = f2_sub2(f2_quad_bcast<3>(m4), f2_quad_bcast<2>(m4));
    r.zz = ZZ3;
    return r;
}

}  // namespace zk
