// curve.cuh -- BLS12-381 G1 (y^2 = x^3 + 4) device arithmetic in extended Jacobian "XYZZ"
// coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2).  XYZZ mixed addition costs 8M + 2S against
// 7M + 4S for the Jacobian madd ark-ec uses (dmsm.rs:23 -> ark-ec 0.4.2 msm): one Fq
// multiplication fewer per bucket update, and Fq mul is the unit everything is priced in.
// Any addition law yields the same affine point, and parity is defined on the affine point.
#pragma once
#include "fp.cuh"

namespace zk {

struct Aff {  // 96 B in HBM: x || y, Montgomery; x = y = 0 encodes infinity (0,0 is not on the curve)
    Fq x, y;
};
struct Xyzz {  // 192 B in HBM; zz == 0 encodes infinity
    Fq x, y, zz, zzz;
};

__device__ __forceinline__ bool aff_is_inf(const Aff& p) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= p.x.l[i] | p.y.l[i];
    return o == 0;
}
__device__ __forceinline__ bool xyzz_is_inf(const Xyzz& p) { return fp_is_zero<FqCfg>(p.zz); }
__device__ __forceinline__ void xyzz_set_inf(Xyzz& p) {
    p.x = fp_one<FqCfg>();
    p.y = fp_one<FqCfg>();
    p.zz = fp_zero<FqCfg>();
    p.zzz = fp_zero<FqCfg>();
}

__device__ __forceinline__ Aff aff_load(const void* base, size_t idx) {
    Aff p;
    const char* b = reinterpret_cast<const char*>(base) + idx * 96;
    p.x = fp_load<FqCfg>(b, 0);
    p.y = fp_load<FqCfg>(b + 48, 0);
    return p;
}
__device__ __forceinline__ Xyzz xyzz_load(const void* base, size_t idx) {
    Xyzz p;
    const char* b = reinterpret_cast<const char*>(base) + idx * 192;
    p.x = fp_load<FqCfg>(b, 0);
    p.y = fp_load<FqCfg>(b + 48, 0);
    p.zz = fp_load<FqCfg>(b + 96, 0);
    p.zzz = fp_load<FqCfg>(b + 144, 0);
    return p;
}
__device__ __forceinline__ void xyzz_store(void* base, size_t idx, const Xyzz& p) {
    char* b = reinterpret_cast<char*>(base) + idx * 192;
    fp_store<FqCfg>(b, 0, p.x);
    fp_store<FqCfg>(b + 48, 0, p.y);
    fp_store<FqCfg>(b + 96, 0, p.zz);
    fp_store<FqCfg>(b + 144, 0, p.zzz);
}

// 2*(x, y) for an affine point, result in XYZZ (mdbl-2008-s-1, a = 0): 3M + 3S... rare path
// (by value + noinline: the rare path stays out of the hot loop's code and never forces the
// caller's accumulator out of registers)
__device__ __noinline__ Xyzz xyzz_dbl_affine(Fq x, Fq y) {
    Xyzz r;
    Fq U = fq_add(y, y);
    Fq V = fq_sqr(U);
    Fq W = fq_mul(U, V);
    Fq S = fq_mul(x, V);
    Fq xx = fq_sqr(x);
    Fq M = fq_add(fq_add(xx, xx), xx);
    Fq X3 = fq_sub(fq_sub(fq_sqr(M), S), S);
    r.y = fq_sub(fq_mul(M, fq_sub(S, X3)), fq_mul(W, y));
    r.x = X3;
    r.zz = V;
    r.zzz = W;
    return r;
}
// 2*p for XYZZ p (dbl-2008-s-1, a = 0)
__device__ __noinline__ Xyzz xyzz_dbl(Xyzz p) {
    Xyzz r;
    if (xyzz_is_inf(p)) return p;
    Fq U = fq_add(p.y, p.y);
    Fq V = fq_sqr(U);
    Fq W = fq_mul(U, V);
    Fq S = fq_mul(p.x, V);
    Fq xx = fq_sqr(p.x);
    Fq M = fq_add(fq_add(xx, xx), xx);
    Fq X3 = fq_sub(fq_sub(fq_sqr(M), S), S);
    Fq Y3 = fq_sub(fq_mul(M, fq_sub(S, X3)), fq_mul(W, p.y));
    r.zz = fq_mul(V, p.zz);
    r.zzz = fq_mul(W, p.zzz);
    r.x = X3;
    r.y = Y3;
    return r;
}

// acc += p (affine), madd-2008-s: 8M + 2S.  `neg` adds -p instead (signed Pippenger digits).
__device__ __forceinline__ void xyzz_madd(Xyzz& acc, const Aff& p_in, bool neg) {
    if (aff_is_inf(p_in)) return;
    Fq py = neg ? fp_neg<FqCfg>(p_in.y) : p_in.y;
    if (xyzz_is_inf(acc)) {
        acc.x = p_in.x;
        acc.y = py;
        acc.zz = fp_one<FqCfg>();
        acc.zzz = fp_one<FqCfg>();
        return;
    }
    Fq U2 = fq_mul(p_in.x, acc.zz);
    Fq S2 = fq_mul(py, acc.zzz);
    Fq P = fq_sub(U2, acc.x);
    Fq R = fq_sub(S2, acc.y);
    if (fp_is_zero<FqCfg>(P)) {  // same x: doubling or cancellation (adversarial inputs only)
        if (fp_is_zero<FqCfg>(R)) acc = xyzz_dbl_affine(p_in.x, py);
        else xyzz_set_inf(acc);
        return;
    }
    Fq PP = fq_sqr(P);
    Fq PPP = fq_mul(P, PP);
    Fq Q = fq_mul(acc.x, PP);
    Fq X3 = fq_sub(fq_sub(fq_sub(fq_sqr(R), PPP), Q), Q);
    Fq Y3 = fq_sub(fq_mul(R, fq_sub(Q, X3)), fq_mul(acc.y, PPP));
    acc.zz = fq_mul(acc.zz, PP);
    acc.zzz = fq_mul(acc.zzz, PPP);
    acc.x = X3;
    acc.y = Y3;
}

// r = a + b, add-2008-s: 12M + 2S
__device__ __forceinline__ Xyzz xyzz_add(const Xyzz& a, const Xyzz& b) {
    if (xyzz_is_inf(a)) return b;
    if (xyzz_is_inf(b)) return a;
    Fq U1 = fq_mul(a.x, b.zz);
    Fq U2 = fq_mul(b.x, a.zz);
    Fq S1 = fq_mul(a.y, b.zzz);
    Fq S2 = fq_mul(b.y, a.zzz);
    Fq P = fq_sub(U2, U1);
    Fq R = fq_sub(S2, S1);
    Xyzz r;
    if (fp_is_zero<FqCfg>(P)) {
        if (fp_is_zero<FqCfg>(R)) r = xyzz_dbl(a);
        else xyzz_set_inf(r);
        return r;
    }
    Fq PP = fq_sqr(P);
    Fq PPP = fq_mul(P, PP);
    Fq Q = fq_mul(U1, PP);
    r.x = fq_sub(fq_sub(fq_sub(fq_sqr(R), PPP), Q), Q);
    r.y = fq_sub(fq_mul(R, fq_sub(Q, r.x)), fq_mul(S1, PPP));
    r.zz = fq_mul(fq_mul(a.zz, b.zz), PP);
    r.zzz = fq_mul(fq_mul(a.zzz, b.zzz), PPP);
    return r;
}

}  // namespace zk
