// curve30.cuh -- G1 XYZZ arithmetic over the unsaturated field (fq30.cuh), with static bounds.
//
// Value bounds (multiples of q) are invariants of the representation, documented per formula:
//   affine SRS points:  x, y < q (canonical, internal Montgomery form 2^390)
//   XYZZ points:        X < 8q, Y < 4q, ZZ < 2q, ZZZ < 2q   (in registers and in HBM; 8q < 2^384)
//   (HBM layout of XYZZ arrays: blocks of 64 points, see xyzz30_load)
// Every product below has input bounds multiplying to <= 256, hence result < 2q (fq30.cuh).
// Infinity = all limbs of ZZ exactly zero (only ever produced by explicit assignment).
#pragma once
#include "fq30.cuh"

namespace zk {

struct Aff30 {
    Fq30 x, y;
};
struct Xyzz30 {
    Fq30 x, y, zz, zzz;
};

__device__ __forceinline__ bool aff30_is_inf(const Aff30& p) { return f30_all_zero(p.x) && f30_all_zero(p.y); }
__device__ __forceinline__ bool xyzz30_is_inf(const Xyzz30& p) { return f30_all_zero(p.zz); }
__device__ __forceinline__ void xyzz30_set_inf(Xyzz30& p) {
    p.x = f30_zero();
    p.y = f30_zero();
    p.zz = f30_zero();
    p.zzz = f30_zero();
}
__device__ __forceinline__ Aff30 aff30_load(const void* base, size_t idx) {
    Aff30 p;
    p.x = f30_load(base, idx * 96);
    p.y = f30_load(base, idx * 96 + 48);
    return p;
}
// XYZZ arrays (buckets, tile heads/tails, reduction rows) are stored in blocks of 64 points, 16-byte
// chunk k (0..11, three per coordinate) of the 64 points side by side: chunk k of point i lives at byte
// (i / 64) * 12288 + k * 1024 + (i % 64) * 16.  Lanes that walk consecutive points -- the fix-up, every
// reduction pass, the head / tail stores of the accumulation -- then move 1 KiB contiguous per load or
// store instruction instead of 64 pieces 192 B apart.  Arrays are allocated in multiples of 64 points.
__device__ __forceinline__ size_t xyzz30_chunk_off(size_t idx, int k) { return (idx >> 6) * 12288 + (size_t)k * 1024 + (idx & 63) * 16; }
__device__ __forceinline__ Fq30 f30_load_chunks(const void* base, size_t idx, int k0) {
    const char* b = reinterpret_cast<const char*>(base);
    u32 w[12];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint4 v = *reinterpret_cast<const uint4*>(b + xyzz30_chunk_off(idx, k0 + i));
        w[4 * i] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
    }
    return f30_from_words(w);
}
__device__ __forceinline__ void f30_store_chunks(void* base, size_t idx, int k0, const Fq30& a) {
    char* b = reinterpret_cast<char*>(base);
    u32 w[12];
    f30_to_words(a, w);
#pragma unroll
    for (int i = 0; i < 3; i++)
        *reinterpret_cast<uint4*>(b + xyzz30_chunk_off(idx, k0 + i)) = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ Xyzz30 xyzz30_load(const void* base, size_t idx) {
    Xyzz30 p;
    p.x = f30_load_chunks(base, idx, 0);
    p.y = f30_load_chunks(base, idx, 3);
    p.zz = f30_load_chunks(base, idx, 6);
    p.zzz = f30_load_chunks(base, idx, 9);
    return p;
}
__device__ __forceinline__ void xyzz30_store(void* base, size_t idx, const Xyzz30& p) {
    f30_store_chunks(base, idx, 0, p.x);
    f30_store_chunks(base, idx, 3, p.y);
    f30_store_chunks(base, idx, 6, p.zz);
    f30_store_chunks(base, idx, 9, p.zzz);
}
// plain 192-byte records (test hook output read by the host)
__device__ __forceinline__ void xyzz30_store_flat(void* base, size_t idx, const Xyzz30& p) {
    f30_store(base, idx * 192, p.x);
    f30_store(base, idx * 192 + 48, p.y);
    f30_store(base, idx * 192 + 96, p.zz);
    f30_store(base, idx * 192 + 144, p.zzz);
}

// q - y for canonical y (0 -> 0): exact negation of an affine coordinate (rare paths only)
__device__ __forceinline__ Fq30 f30_neg_canon(const Fq30& y) {
    if (f30_all_zero(y)) return y;
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = Q30::KQ2(i) - y.l[i];  // 2q - y in (q, 2q)
    f30_norm(r);
    return f30_csub_q(r);
}

// 2*(x, y) for an affine point, x < 8q, y < 4q (mdbl-2008-s-1, a = 0).  Rare path.
__device__ __noinline__ Xyzz30 xyzz30_dbl_affine(Fq30 x, Fq30 y) {
    Xyzz30 r;
    Fq30 U = f30_add(y, y);                   // < 8q
    Fq30 V = f30_sqr(U);                      // < 2q
    Fq30 W = f30_mul(U, V);                   // < 2q
    Fq30 S = f30_mul(x, V);                   // < 2q
    Fq30 xx = f30_sqr(x);                     // < 2q
    Fq30 M = f30_add(f30_add(xx, xx), xx);    // < 6q
    Fq30 S2 = f30_add(S, S);                  // < 4q
    r.x = f30_sub4(f30_sqr(M), S2);           // M^2 + 4q - 2S < 6q
    r.y = f30_sub2(f30_mul(M, f30_sub8(S, r.x)), f30_mul(W, y));  // M(S + 8q - X3) + 2q - W*y < 4q
    r.zz = V;
    r.zzz = W;
    return r;
}
// 2*p for XYZZ p (dbl-2008-s-1, a = 0).  Rare path.
__device__ __noinline__ Xyzz30 xyzz30_dbl(Xyzz30 p) {
    if (xyzz30_is_inf(p)) return p;
    Xyzz30 r;
    Fq30 U = f30_add(p.y, p.y);               // < 8q
    Fq30 V = f30_sqr(U);
    Fq30 W = f30_mul(U, V);
    Fq30 S = f30_mul(p.x, V);                 // 8q * 2q
    Fq30 xx = f30_sqr(p.x);                   // 64 q^2
    Fq30 M = f30_add(f30_add(xx, xx), xx);    // < 6q
    Fq30 S2 = f30_add(S, S);
    Fq30 X3 = f30_sub4(f30_sqr(M), S2);       // < 6q
    r.y = f30_sub2(f30_mul(M, f30_sub8(S, X3)), f30_mul(W, p.y));
    r.x = X3;
    r.zz = f30_mul(V, p.zz);
    r.zzz = f30_mul(W, p.zzz);
    return r;
}

// acc += p (affine; `neg`: -p), madd-2008-s: 8M + 2S, no modular reduction on the hot path.
// The doubling / cancellation case (same x) is detected AFTER the fact: then P == 0 (mod q), so
// ZZ3 = ZZ1 * P^2 == 0 (mod q), which is a two-constant comparison on a value < 2q.
__device__ __forceinline__ void xyzz30_madd(Xyzz30& acc, const Aff30& p, bool neg) {
    if (aff30_is_inf(p)) return;
    if (xyzz30_is_inf(acc)) {
        acc.x = p.x;
        acc.y = neg ? f30_neg_canon(p.y) : p.y;
        acc.zz = f30_one();
        acc.zzz = f30_one();
        return;
    }
    const Fq30 U2 = f30_mul(p.x, acc.zz);     // q * 2q -> < 2q
    const Fq30 S2 = f30_mul(p.y, acc.zzz);    // < 2q
    const Fq30 P = f30_sub8(U2, acc.x);       // U2 + 8q - X1 < 10q
    // R = +-S2 - Y1:  S2 + 4q - Y1 < 6q   or   (4q - Y1) + 2q - S2 < 6q
    Fq30 R;
#pragma unroll
    for (int i = 0; i < 13; i++)  // one limb-wise pass for both signs: (KQ2 - S2 | S2) + KQ4 - Y1, every limb stays below 2^32
        R.l[i] = (neg ? Q30::KQ2(i) + Q30::KQ4(i) - S2.l[i] : S2.l[i] + Q30::KQ4(i)) - acc.y.l[i];
    f30_norm(R);
    const Fq30 PP = f30_sqr(P);               // 100 q^2 -> < 2q
    const Fq30 PPP = f30_mul(P, PP);          // < 2q
    const Fq30 Q = f30_mul(acc.x, PP);        // 8q * 2q -> < 2q
    const Fq30 ZZ3 = f30_mul(acc.zz, PP);     // < 2q
    if (f30_is_zero_2q(ZZ3)) {                // same x coordinate: doubling or cancellation (adversarial inputs only)
        const Fq30 Rc = f30_canon8(R);
        if (f30_all_zero(Rc)) acc = xyzz30_dbl_affine(p.x, neg ? f30_neg_canon(p.y) : p.y);
        else xyzz30_set_inf(acc);
        return;
    }
    const Fq30 X3 = f30_sub6(f30_sqr(R), f30_add2x(PPP, Q));    // R^2 + 6q - (PPP + 2Q) < 8q
    // R(Q + 8q - X3) + (4q - Y1)*PPP under one reduction: 6q*10q + 4q*2q <= 256 q^2 -> < 2q
    const Fq30 Y3 = f30_mul2add(R, f30_sub8(Q, X3), f30_sub4(f30_zero(), acc.y), PPP);
    acc.zzz = f30_mul(acc.zzz, PPP);
    acc.zz = ZZ3;
    acc.x = X3;
    acc.y = Y3;
}

// r = a + b (XYZZ + XYZZ), add-2008-s: 12M + 2S
__device__ __forceinline__ Xyzz30 xyzz30_add(const Xyzz30& a, const Xyzz30& b) {
    if (xyzz30_is_inf(a)) return b;
    if (xyzz30_is_inf(b)) return a;
    const Fq30 U1 = f30_mul(a.x, b.zz);       // 8q * 2q
    const Fq30 U2 = f30_mul(b.x, a.zz);
    const Fq30 S1 = f30_mul(a.y, b.zzz);      // 4q * 2q
    const Fq30 S2 = f30_mul(b.y, a.zzz);
    const Fq30 P = f30_sub2(U2, U1);          // < 4q
    const Fq30 R = f30_sub2(S2, S1);          // < 4q
    const Fq30 PP = f30_sqr(P);
    const Fq30 PPP = f30_mul(P, PP);
    const Fq30 Q = f30_mul(U1, PP);
    const Fq30 ZZ3 = f30_mul(f30_mul(a.zz, b.zz), PP);
    Xyzz30 r;
    if (f30_is_zero_2q(ZZ3)) {
        if (f30_all_zero(f30_canon8(R))) r = xyzz30_dbl(a);
        else xyzz30_set_inf(r);
        return r;
    }
    r.x = f30_sub6(f30_sqr(R), f30_add2x(PPP, Q));                          // R^2 + 6q - (PPP + 2Q) < 8q
    r.y = f30_mul2add(R, f30_sub8(Q, r.x), f30_sub2(f30_zero(), S1), PPP);  // 4q*10q + 2q*2q -> < 2q
    r.zz = ZZ3;
    r.zzz = f30_mul(f30_mul(a.zzz, b.zzz), PPP);
    return r;
}

// ---- one XYZZ addition spread over the four lanes of a quad (latency-bound reduction passes) ----
// The 13 field multiplications of add-2008-s have depth 4; a quad runs them as 4 rounds of one
// multiplication per lane, exchanging operands with DPP quad broadcasts (no LDS, no memory).
template <int R>
__device__ __forceinline__ Fq30 f30_quad_bcast(const Fq30& v) {  // lane R's value to the 4 lanes of its quad
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)v.l[i], R * 0x55, 0xf, 0xf, true);
        // keep the broadcast a plain v_mov_b32_dpp: folded into a following subtraction (DPP combine) the
        // lane permutation landed on the wrong operand (observed: r.y of xyzz30_acc_quad wrong in 3 of 4 lanes)
        asm volatile("" : "+v"(r.l[i]));
    }
    return r;
}
__device__ __forceinline__ Fq30 f30_sel4(int role, const Fq30& a0, const Fq30& a1, const Fq30& a2, const Fq30& a3) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = (role == 0) ? a0.l[i] : (role == 1) ? a1.l[i] : (role == 2) ? a2.l[i] : a3.l[i];
    return r;
}
// out[io] = in[ia] + in[ib]; called by all four lanes of a quad (role = lane & 3) with identical indices.
// Round 1: U1 = X1*ZZ2 | U2 = X2*ZZ1 | S1 = Y1*ZZZ2 | S2 = Y2*ZZZ1      (lane 0 | 1 | 2 | 3)
// Round 2: A = ZZ1*ZZ2 | B = ZZZ1*ZZZ2 | PP = P^2 | RR = R^2
// Round 3: Q = U1*PP   | ZZ3 = A*PP    | PPP = P*PP | -
// Round 4: -           | ZZZ3 = B*PPP  | T1 = S1*PPP | T2 = R*(Q + 8q - X3);  Y3 = T2 - T1
__device__ __forceinline__ void xyzz30_add_quad(const void* __restrict__ in, size_t ia, size_t ib, void* __restrict__ out, size_t io,
                                               int role) {
    const Fq30 opA = f30_load_chunks(in, (role & 1) ? ib : ia, (role & 2) ? 3 : 0);  // a.x | b.x | a.y | b.y
    const Fq30 opB = f30_load_chunks(in, (role & 1) ? ia : ib, (role & 2) ? 9 : 6);  // b.zz | a.zz | b.zzz | a.zzz
    const Fq30 zz2 = f30_quad_bcast<0>(opB), zz1 = f30_quad_bcast<1>(opB), zzz2 = f30_quad_bcast<2>(opB), zzz1 = f30_quad_bcast<3>(opB);
    const bool a_inf = f30_all_zero(zz1), b_inf = f30_all_zero(zz2);  // identical in the four lanes
    if (a_inf || b_inf) {  // the other operand (or infinity) is the result: lane r copies coordinate r
        f30_store_chunks(out, io, 3 * role, f30_load_chunks(in, a_inf ? ib : ia, 3 * role));
        return;
    }
    const Fq30 m1 = f30_mul(opA, opB);
    const Fq30 u1 = f30_quad_bcast<0>(m1), u2 = f30_quad_bcast<1>(m1), s1 = f30_quad_bcast<2>(m1), s2 = f30_quad_bcast<3>(m1);
    const Fq30 P = f30_sub2(u2, u1);  // < 4q
    const Fq30 R = f30_sub2(s2, s1);  // < 4q
    const Fq30 m2 = f30_mul(f30_sel4(role, zz1, zzz1, P, R), f30_sel4(role, zz2, zzz2, P, R));
    const Fq30 A = f30_quad_bcast<0>(m2), B = f30_quad_bcast<1>(m2), PP = f30_quad_bcast<2>(m2), RR = f30_quad_bcast<3>(m2);
    const Fq30 m3 = f30_mul(f30_sel4(role, u1, A, P, P), PP);
    const Fq30 Q = f30_quad_bcast<0>(m3), ZZ3 = f30_quad_bcast<1>(m3), PPP = f30_quad_bcast<2>(m3);
    if (f30_is_zero_2q(ZZ3)) {  // same x: doubling or cancellation (adversarial inputs only) -- lane 0 redoes it alone
        if (role == 0) xyzz30_store(out, io, xyzz30_add(xyzz30_load(in, ia), xyzz30_load(in, ib)));
        return;
    }
    const Fq30 X3 = f30_sub6(RR, f30_add2x(PPP, Q));  // < 8q
    const Fq30 m4 = f30_mul(f30_sel4(role, B, B, s1, R), f30_sel4(role, PPP, PPP, PPP, f30_sub8(Q, X3)));
    const Fq30 ZZZ3 = f30_quad_bcast<1>(m4), T1 = f30_quad_bcast<2>(m4), T2 = f30_quad_bcast<3>(m4);
    const Fq30 Y3 = f30_sub2(T2, T1);  // < 4q
    f30_store_chunks(out, io, 3 * role, f30_sel4(role, X3, Y3, ZZ3, ZZZ3));
}

// acc += in[ib] with acc held (replicated) in the registers of all four lanes of the quad: the chained
// form for the fix-up, same rounds as xyzz30_add_quad.  Each lane fetches the one coordinate of b its
// round-1 product needs (b.zz | b.x | b.zzz | b.y).
__device__ __forceinline__ Xyzz30 xyzz30_acc_quad(const Xyzz30& acc, const void* __restrict__ in, size_t ib, int role) {
    const Fq30 bpart = f30_load_chunks(in, ib, role == 0 ? 6 : role == 1 ? 0 : role == 2 ? 9 : 3);
    const Fq30 zz2 = f30_quad_bcast<0>(bpart);
    if (f30_all_zero(zz2)) return acc;                       // b is infinity
    if (xyzz30_is_inf(acc)) return xyzz30_load(in, ib);      // (all lanes load all of b)
    const Fq30 zzz2 = f30_quad_bcast<2>(bpart);
    const Fq30 m1 = f30_mul(f30_sel4(role, acc.x, bpart, acc.y, bpart), f30_sel4(role, bpart, acc.zz, bpart, acc.zzz));
    const Fq30 u1 = f30_quad_bcast<0>(m1), u2 = f30_quad_bcast<1>(m1), s1 = f30_quad_bcast<2>(m1), s2 = f30_quad_bcast<3>(m1);
    const Fq30 P = f30_sub2(u2, u1);  // < 4q
    const Fq30 R = f30_sub2(s2, s1);  // < 4q
    const Fq30 m2 = f30_mul(f30_sel4(role, acc.zz, acc.zzz, P, R), f30_sel4(role, zz2, zzz2, P, R));
    const Fq30 A = f30_quad_bcast<0>(m2), B = f30_quad_bcast<1>(m2), PP = f30_quad_bcast<2>(m2), RR = f30_quad_bcast<3>(m2);
    const Fq30 m3 = f30_mul(f30_sel4(role, u1, A, P, P), PP);
    const Fq30 Q = f30_quad_bcast<0>(m3), ZZ3 = f30_quad_bcast<1>(m3), PPP = f30_quad_bcast<2>(m3);
    if (f30_is_zero_2q(ZZ3)) return xyzz30_add(acc, xyzz30_load(in, ib));  // doubling / cancellation: every lane alone, same result
    Xyzz30 r;
    r.x = f30_sub6(RR, f30_add2x(PPP, Q));  // < 8q
    const Fq30 m4 = f30_mul(f30_sel4(role, B, B, s1, R), f30_sel4(role, PPP, PPP, PPP, f30_sub8(Q, r.x)));
    r.zzz = f30_quad_bcast<1>(m4);
    r.y = f30_sub2(f30_quad_bcast<3>(m4), f30_quad_bcast<2>(m4));  // < 4q
    r.zz = ZZ3;
    return r;
}

}  // namespace zk
