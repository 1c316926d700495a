// zk_msm.hip -- G1 multi-scalar multiplication on gfx950: the `G::msm(bases, scalars)` call of
// d_msm (dist-primitive/src/dmsm.rs:23) and of commit/open (dpoly_comm.rs:242,274,457).
//
// Pipeline (signed-digit Pippenger over the 2n points P_i, phi(P_i) with 128-bit scalar halves;
// window c, W = ceil(129 / c) windows, nb = 2^(c-1) buckets per window):
//   1 k_digits        scalars out of Montgomery form (as ark `into_bigint`), GLV split k = k1 + k2*lambda,
//                     signed digits of both halves                            [coalesced 32-B reads]
//   2 k_part_hist, k_part_scan, k_part_bases, k_part_scatter, k_part_sort
//                     two-level counting sort, counters in LDS: entries grouped by bucket
//   3 k_accum_tiles   every lane walks T consecutive sorted entries, gathers 96-B affine bases from
//                     HBM and accumulates with XYZZ mixed adds (8M+2S)        [the dominant kernel]
//   4 k_fixup(_long)  stitches the buckets cut by tile boundaries
//   5 k_halve         bucket reduction WITHOUT the serial running sum: sum_b b*B_b is split into
//                     bit planes.  Each pass pairs neighbours (L[2j]+L[2j+1]) and peels the odd
//                     elements off as a new row whose plain sum is the plane T_k; rows keep halving.
//                     After c-1 passes every window is down to c points (T_0..T_{c-2}, T_all).
//   6 k_finish        those points as Jacobian coordinates in the reference Montgomery form
//   7 host            sum_w 2^{o_w} (T_all + sum_k 2^k T_k): ~129 doublings, a pure dependency chain
//                     (host_curve.hpp), then normalisation to affine.
// Field arithmetic: csrc/fq30.cuh (13 x 30-bit limbs, Montgomery radix 2^390, lazy reduction).
// All additions are exact group operations, so the affine result is independent of the
// (non-deterministic) order in which the sort places points inside a bucket.
#include "curve30.cuh"
#include "curve30_g2.cuh"
#include "host_curve.hpp"
#include "zk_ctx.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// This file is compiled TWICE (csrc/Makefile): once as it is -- the G1 pipeline and every non-templated entry point -- and once with
// -DZK_MSM_TU_G2, which keeps only what instantiates the templates over CvG2 (msm_g2_batch, the G2 window-table kernel, the G2
// test hook).  The two translation units build in parallel (the Fq2 kernels are more than half of the compile time); kernels that
// are not templates are `static`, so each unit carries its own copy of the few it launches.
namespace zk {

static constexpr int kBlk = 256;
static constexpr u32 kSkip = 0xffffffffu;


// ---------------------------------------------------------------------------------------
// The pipeline is generic over the curve (`d_msm<G: CurveGroup>`, dmsm.rs:9): G1 over Fq (curve30.cuh) and G2
// over Fq2 (curve30_g2.cuh).  A traits struct names the point types and operations; for G1 it forwards to the
// functions the kernels always used (identical code), G2 instantiates the same kernels a second time.
// ---------------------------------------------------------------------------------------
struct CvG1 {
    typedef Aff30 Aff;
    typedef Xyzz30 Xyzz;
    typedef zkhost::Fq HostF;
    static constexpr size_t kAffBytes = 96, kXyzzBytes = 192, kJacBytes = 144;
    static constexpr bool kEndo = true;  // scalars split with the endomorphism, SRS holds phi(P_i) after P_i
    static constexpr bool kQuad = true;  // quad-lane fix-up / reduction kernels exist
    __device__ static __forceinline__ Aff aff_load(const void* b, size_t i) { return aff30_load(b, i); }
    __device__ static __forceinline__ Aff aff_load_rec(const void* b, size_t i, u32 rec) {
        Aff p;
        p.x = f30_load(b, i * rec);
        p.y = f30_load(b, i * rec + 48);
        return p;
    }
    __device__ static __forceinline__ Xyzz load(const void* b, size_t i) { return xyzz30_load(b, i); }
    __device__ static __forceinline__ void store(void* b, size_t i, const Xyzz& p) { xyzz30_store(b, i, p); }
    __device__ static __forceinline__ void set_inf(Xyzz& p) { xyzz30_set_inf(p); }
    __device__ static __forceinline__ bool is_inf(const Xyzz& p) { return xyzz30_is_inf(p); }
    __device__ static __forceinline__ void madd(Xyzz& acc, const Aff& p, bool neg) { xyzz30_madd(acc, p, neg); }
    __device__ static __forceinline__ Xyzz add(const Xyzz& a, const Xyzz& b) { return xyzz30_add(a, b); }
    __device__ static __forceinline__ Xyzz dbl(const Xyzz& a) { return xyzz30_dbl(a); }
    // quad-lane forms (role = lane & 3 owns coordinate `role` of a point)
    __device__ static __forceinline__ void quad_copy(const void* in, size_t src, void* out, size_t dst, int role) {
        f30_store_chunks(out, dst, 3 * role, f30_load_chunks(in, src, 3 * role));
    }
    __device__ static __forceinline__ void quad_store_inf(void* out, size_t dst, int role) { f30_store_chunks(out, dst, 3 * role, f30_zero()); }
    __device__ static __forceinline__ void quad_store(void* out, size_t dst, int role, const Xyzz& p) {
        f30_store_chunks(out, dst, 3 * role, role == 0 ? p.x : role == 1 ? p.y : role == 2 ? p.zz : p.zzz);
    }
    __device__ static __forceinline__ void add_quad(const void* in, size_t ia, size_t ib, void* out, size_t io, int role) { xyzz30_add_quad(in, ia, ib, out, io, role); }
    __device__ static __forceinline__ Xyzz acc_quad(const Xyzz& acc, const void* in, size_t ib, int role) { return xyzz30_acc_quad(acc, in, ib, role); }
    // the affine image of acc as one record of the window table (canonical coordinates, like every SRS coordinate)
    __device__ static __forceinline__ void table_store(const Xyzz& acc, void* table, size_t idx, u32 rec) {
        Aff30 a;
        if (xyzz30_is_inf(acc)) {
            a.x = f30_zero();
            a.y = f30_zero();
        } else {
            const Fq30 i3 = f30_inv(acc.zzz);               // 1/Z^3
            const Fq30 iz = f30_mul(acc.zz, i3);            // Z^2/Z^3 = 1/Z
            a.x = f30_canon8(f30_mul(acc.x, f30_sqr(iz)));  // X/Z^2
            a.y = f30_canon8(f30_mul(acc.y, i3));           // Y/Z^3
        }
        f30_store(table, idx * rec, a.x);
        f30_store(table, idx * rec + 48, a.y);
    }
    // (X*ZZ, Y*ZZZ, ZZ) is a Jacobian representative; written in the REFERENCE Montgomery form
    __device__ static __forceinline__ void finish(const Xyzz& acc, void* out, size_t t) {
        Fq30 X = f30_zero(), Y = f30_zero(), Z = f30_zero();
        if (!xyzz30_is_inf(acc)) {
            X = f30_to_ref(f30_mul(acc.x, acc.zz));
            Y = f30_to_ref(f30_mul(acc.y, acc.zzz));
            Z = f30_to_ref(acc.zz);
        }
        f30_store(out, t * 144, X);
        f30_store(out, t * 144 + 48, Y);
        f30_store(out, t * 144 + 96, Z);
    }
};
struct CvG2 {
    typedef Aff2 Aff;
    typedef Xyzz2 Xyzz;
    typedef zkhost::Fq2 HostF;
    static constexpr size_t kAffBytes = 192, kXyzzBytes = 384, kJacBytes = 288;
    static constexpr bool kEndo = false;
    static constexpr bool kQuad = true;
    __device__ static __forceinline__ Aff aff_load(const void* b, size_t i) { return aff2_load(b, i); }
    __device__ static __forceinline__ Aff aff_load_rec(const void* b, size_t i, u32 /*rec*/) { return aff_load(b, i); }
    __device__ static __forceinline__ Xyzz load(const void* b, size_t i) { return xyzz2_load(b, i); }
    __device__ static __forceinline__ void store(void* b, size_t i, const Xyzz& p) { xyzz2_store(b, i, p); }
    __device__ static __forceinline__ void set_inf(Xyzz& p) { xyzz2_set_inf(p); }
    __device__ static __forceinline__ bool is_inf(const Xyzz& p) { return xyzz2_is_inf(p); }
    __device__ static __forceinline__ void madd(Xyzz& acc, const Aff& p, bool neg) { xyzz2_madd(acc, p, neg); }
    __device__ static __forceinline__ Xyzz add(const Xyzz& a, const Xyzz& b) { return xyzz2_add(a, b); }
    __device__ static __forceinline__ Xyzz dbl(const Xyzz& a) { return xyzz2_dbl(a); }
    __device__ static __forceinline__ void quad_copy(const void* in, size_t src, void* out, size_t dst, int role) {
        f2_store_chunks(out, dst, 6 * role, f2_load_chunks(in, src, 6 * role));
    }
    __device__ static __forceinline__ void quad_store_inf(void* out, size_t dst, int role) { f2_store_chunks(out, dst, 6 * role, f2_zero()); }
    __device__ static __forceinline__ void quad_store(void* out, size_t dst, int role, const Xyzz& p) {
        f2_store_chunks(out, dst, 6 * role, role == 0 ? p.x : role == 1 ? p.y : role == 2 ? p.zz : p.zzz);
    }
    __device__ static __forceinline__ void add_quad(const void* in, size_t ia, size_t ib, void* out, size_t io, int role) { xyzz2_add_quad(in, ia, ib, out, io, role); }
    __device__ static __forceinline__ Xyzz acc_quad(const Xyzz& acc, const void* in, size_t ib, int role) { return xyzz2_acc_quad(acc, in, ib, role); }
    __device__ static __forceinline__ void table_store(const Xyzz& acc, void* table, size_t idx, u32 /*rec: always kAffBytes*/) {
        Fq2x x = f2_zero(), y = f2_zero();
        if (!xyzz2_is_inf(acc)) {
            // 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2)
            const Fq30 nrm = f30_mul2add(acc.zzz.c0, acc.zzz.c0, acc.zzz.c1, acc.zzz.c1);  // 2q * 2q * 2 <= 256: < 2q
            const Fq30 ni = f30_inv(nrm);
            const Fq2x i3 = Fq2x{f30_mul(acc.zzz.c0, ni), f30_mul(f30_negk<2>(acc.zzz.c1), ni)};  // 1/Z^3, components < 2q
            const Fq2x iz = f2_mul<2>(acc.zz, i3);                                                    // 1/Z
            x = f2_canon8(f2_mul<2>(acc.x, f2_sqr<2>(iz)));                                           // 8 * 4
            y = f2_canon8(f2_mul<2>(acc.y, i3));
        }
        f30_store(table, idx * 192, x.c0);
        f30_store(table, idx * 192 + 48, x.c1);
        f30_store(table, idx * 192 + 96, y.c0);
        f30_store(table, idx * 192 + 144, y.c1);
    }
    __device__ static __forceinline__ void finish(const Xyzz& acc, void* out, size_t t) {
        Fq2x X = f2_zero(), Y = f2_zero(), Z = f2_zero();
        if (!xyzz2_is_inf(acc)) {
            X = f2_mul<2>(acc.x, acc.zz);
            Y = f2_mul<2>(acc.y, acc.zzz);
            Z = acc.zz;
        }
        const Fq30 c[6] = {X.c0, X.c1, Y.c0, Y.c1, Z.c0, Z.c1};
#pragma unroll
        for (int k = 0; k < 6; k++) f30_store(out, t * 288 + 48 * k, f30_to_ref(c[k]));
    }
};

// The scalar is split with the curve endomorphism (see k_digits): k = k1 + k2*lambda, k1, k2 < 2^128,
// and the MSM runs over the 2n points P_i, phi(P_i) with 128-bit scalars: half the windows, hence half
// the buckets to reduce and half the host doubling chain, for the same number of bucket additions.
static constexpr int kEndoBits = 129;  // 128 bits + room for the carry of the signed recoding
static constexpr int kFullBits = 256;  // 255-bit scalar + carry (precomputed-table mode)
static constexpr int kMaxTableBits = 22;  // widest window of a table: 2^21 buckets = kMaxParts partitions x kMaxLow buckets of the sort

// window bits for an n-point MSM (2n entries per window after the split).  Up to 2^21 points the
// measured optimum follows log2(n) - 3, log2(n) - 2 below 2^16 (sweeps in tools/sweep_msm.py); from 2^22 on 19 bits (7 windows
// instead of 8) saves more bucket additions (10 Fq-mul each, 2n per window) than the ~4x larger bucket
// reduction (3 x 14 Fq-mul per bucket) costs.
#ifndef ZK_MSM_TU_G2  // (G1 translation unit only)
int msm_pick_window(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    if (lg >= 22) return 19;
    int c = lg <= 15 ? lg - 2 : lg - 3;  // small MSMs are latency chains: fewer, wider windows
    if (c < 5) c = 5;
    if (c > 17) c = 17;
    return c;
}
#endif
// 256-bit layout of the precomputed-table mode: ONE bucket set for all ceil(256 / c) digits of a scalar, so the window
// can be as wide as the bucket reduction of 2^(c-1) buckets allows.  Measured optimum (tools/sweep_pre.py, MI355X):
// log2 n + 2 up to 2^15 points, 17 bits for 2^16..2^18 (one more bit doubles the narrow-row fix-up), log2 n - 1 from 2^19.
static int msm_pick_window_full(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = lg <= 15 ? lg + 2 : (lg <= 18 ? 17 : lg - 1);
    // two common widths for the short levels: in a batch (a proof's passes hold ~11 items of every size 2^6 .. 2^14) items of one
    // table width and up to 2^msm_size_class_min points form ONE class, i.e. two launch chains instead of nine at the end of
    // every pass; a single short MSM is a latency chain either way (tools/msm_time.py: within 5 %)
    if (tuning().msm_small_table_widths != 0 && lg <= 14) c = lg <= 10 ? 12 : 14;
    c += (int)tuning().msm_table_dc;  // (sweeps)
    if (c < 4) c = 4;
    if (c > 20) c = 20;  // (the automatic pick; zk_srs_precompute accepts up to kMaxTableBits -- see the sweep in profiles/r06h_table_width_sweep.txt)
    return c;
}

// Window layout: the `bits` scalar bits (including room for the last carry) are split into W
// windows whose widths differ by at most one bit (the first `rem` windows are base+1 wide, the
// rest base wide, base+1 <= c).  Balanced widths matter: a narrow (or carry-only) top window would
// funnel ~all points of that window into a handful of buckets.
struct WinLayout {
    int W, base, rem;
    __host__ __device__ int width(int w) const { return base + (w < rem ? 1 : 0); }
    __host__ __device__ int bit_offset(int w) const { return w * base + (w < rem ? w : rem); }
};
static WinLayout msm_layout(int c, int bits) {
    WinLayout L;
    L.W = (bits + c - 1) / c;
    L.base = bits / L.W;
    L.rem = bits % L.W;
    return L;
}

// ---------------------------------------------------------------------------------------
// 1. digits: scalars out of Montgomery form, signed c-bit digits.  digits row stride = ns.
// ---------------------------------------------------------------------------------------
// one MSM of a batch (all items of a launch share the window layout and the row stride ns)
struct ItemDesc {
    const void* scalars;
    const void* bases;  // packed 96-B affine points, already offset
    u32 n;
    u32 pstride;  // a row holds several copies of the point range: entry v -> bases[(v / ns) * pstride + v % ns]
                  // (copy 1 = the endomorphism images phi(P_i); precomputed table: copy w = 2^{offset(w)} P_i)
    u32 rec;      // bytes between two records of `bases` (kAffBytes; a G1 window table may hold 128-B aligned records: zk_srs::table_rec)
};

// GLV split for BLS12-381 G1: lambda = z^2 - 1 (z the curve parameter) satisfies lambda^2 + lambda + 1 = 0
// (mod r) and phi(x, y) = (beta*x, y) = lambda*(x, y) on the r-torsion.  k2 = floor(k / lambda) <= lambda + 1,
// k1 = k - k2*lambda < lambda < 2^128: both halves are non-negative, no lattice rounding needed.
// The quotient comes from a reciprocal (mu = floor(2^256 / lambda)) on the top 159 bits, short by at
// most 2, fixed by comparing the remainder with lambda.
struct Glv {
    static constexpr u32 LAM(int i) {
        constexpr u32 t[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};
        return t[i];
    }
    static constexpr u32 MU(int i) {
        constexpr u32 t[5] = {0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u};
        return t[i];
    }
};
// k (8 limbs, < r) -> k1 (5 limbs, top limb 0), k2 (5 limbs, top limb 0)
__device__ __forceinline__ void glv_split(const Fr& k, u32 (&k1)[5], u32 (&k2)[5]) {
    // qh = ((k >> 96) * mu) >> 160
    u32 prod[10];
#pragma unroll
    for (int i = 0; i < 10; i++) prod[i] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            u64 t = (u64)k.l[3 + i] * Glv::MU(j) + prod[i + j] + carry;
            prod[i + j] = (u32)t;
            carry = t >> 32;
        }
        prod[i + 5] = (u32)carry;
    }
    u32 q[5] = {prod[5], prod[6], prod[7], prod[8], 0};
    // rem = k - q * lambda  (fits 130 bits; computed mod 2^160)
    u32 ql[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i + j < 5) {
                u64 t = (u64)q[i] * Glv::LAM(j) + ql[i + j] + carry;
                ql[i + j] = (u32)t;
                carry = t >> 32;
            }
        }
        if (i + 4 < 5) ql[i + 4] += (u32)carry;
    }
    u32 rem[5];
    {
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            u64 t = (u64)k.l[i] - ql[i] - bw;
            rem[i] = (u32)t;
            bw = (u32)(t >> 63);
        }
    }
    for (int it = 0; it < 3; it++) {  // remainder >= lambda: one more lambda goes into the quotient
        u32 d[5], bw = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            u64 t = (u64)rem[i] - (i < 4 ? Glv::LAM(i) : 0u) - bw;
            d[i] = (u32)t;
            bw = (u32)(t >> 63);
        }
        if (bw) break;
#pragma unroll
        for (int i = 0; i < 5; i++) rem[i] = d[i];
        u32 c = 1;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            u32 v = q[i] + c;
            c = (v < c) ? 1u : 0u;
            q[i] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        k1[i] = rem[i];
        k2[i] = q[i];
    }
}

// signed recoding of one (<= 8-limb) value over the windows [0, w0 + wc) of the layout; rows of
// the windows >= w0 are written at digits[(w - w0) * row_len + pos]
template <int NL>
__device__ __forceinline__ void recode_windows(u32 (&s)[NL], const WinLayout& L, int w0, int wc, u32* __restrict__ digits,
                                               size_t row_len, size_t pos) {
    u32 carry = 0;
    for (int w = 0; w < w0 + wc; w++) {
        const int cw = L.width(w);  // <= kMaxTableBits
        const u32 mask = (1u << cw) - 1u, half = 1u << (cw - 1);
        u32 v = (s[0] & mask) + carry;
        // s >>= cw
#pragma unroll
        for (int k = 0; k < NL - 1; k++) s[k] = (s[k] >> cw) | (s[k + 1] << (32 - cw));
        s[NL - 1] >>= cw;
        u32 d;
        if (v > half) {  // recentre to [-2^(cw-1), 2^(cw-1)]; never triggers in the top window
            v = (1u << cw) - v;
            carry = 1;
            d = 0x80000000u;
        } else {
            carry = 0;
            d = 0;
        }
        d = (v == 0) ? kSkip : (d | (v - 1));
        if (w >= w0) digits[(size_t)(w - w0) * row_len + pos] = d;
    }
}

// rows are written for the windows [w0, w0 + wc) only (a class may cover a sub-range of the windows);
// the lower windows are still walked for the carry of the signed recoding.
// endo: row = [k1 digits of the ns points | k2 digits of the ns points] (entry ns + i refers to phi(P_i)).
static __global__ void __launch_bounds__(kBlk) k_digits(const ItemDesc* __restrict__ items, size_t ns, WinLayout L, int w0, int wc, int endo,
                                               u32* __restrict__ digits_all) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= ns) return;
    const ItemDesc it = items[blockIdx.y];
    const size_t row_len = endo ? 2 * ns : ns;
    u32* digits = digits_all + (size_t)blockIdx.y * wc * row_len;
    if (i >= it.n) {  // padding: only up to the item's own length rounded up to 4 -- the sort never reads beyond it (struct RowReal)
        if (i >= (((size_t)it.n + 3) & ~(size_t)3)) return;
        for (int w = 0; w < wc; w++) {
            digits[(size_t)w * row_len + i] = kSkip;
            if (endo) digits[(size_t)w * row_len + ns + i] = kSkip;
        }
        return;
    }
    Fr s = fp_from_mont<FrCfg>(fr_load(it.scalars, i));
    if (endo) {
        u32 k1[5], k2[5];
        glv_split(s, k1, k2);
        recode_windows<5>(k1, L, w0, wc, digits, row_len, i);
        recode_windows<5>(k2, L, w0, wc, digits, row_len, ns + i);
    } else {
        recode_windows<8>(s.l, L, w0, wc, digits, row_len, i);
    }
}

// ---------------------------------------------------------------------------------------
// 2./3. two-level counting sort of the entries of every row (a "row" is one bucket set: a window,
// or -- with a precomputed SRS -- all windows at once) by bucket.  No global atomics; every store
// stream is either coalesced or confined to a region one workgroup owns (so partial lines merge in
// that workgroup's L2 instead of bouncing between XCDs):
//   level 1  the row is cut into chunks; the top bits of the bucket pick one of np = 256..1024 partitions
//     k_part_hist     block (chunk, row): partition histogram in LDS      -> hist[row][p][chunk]
//     k_part_scan     block (p, row): exclusive scan over the chunks      -> hist in place, total[row][p]
//     k_part_bases    block (row): exclusive scan over the partitions     -> base[row][p], rowtot[row]
//     k_part_scatter  block (chunk, row): cursors in LDS; entry (index in row | sign bit) and its
//                     bucket-in-partition go to part_idx / part_low: np streams of ~chunk/np entries
//   level 2  block (p, row) owns its partition: counts the <= 2048 buckets in LDS, scans, writes
//     k_part_sort     counts[row][b], offsets[row][b] and the entries grouped by bucket into sorted[]
// ---------------------------------------------------------------------------------------
static constexpr int kSortThreads = 1024;
static constexpr u32 kMaxParts = 1024;
static constexpr u32 kMaxLow = 2048;  // buckets per partition (nb / np), nb <= 2^19

// A row is `copies` segments of ns slots (ns = the class-wide stride: the longest item, a multiple of 4); an item of n points
// fills the first n slots of every segment.  The sort walks a row in uint4 steps and skips the steps that lie in the padding of
// their segment without touching memory: a class may then hold items of very different lengths (one launch chain for all
// window-table items of one table width) at the price of idle loop iterations, not of traffic.
struct RowReal {
    const ItemDesc* items;
    u32 rpi, ns;
    // slots [4 i4, 4 i4 + 4) of `row`: n4 = the item's length rounded up to 4; pos = offset of the step inside its segment
    __device__ __forceinline__ u32 n4(u32 row) const { return (items[row / rpi].n + 3u) & ~3u; }
    __device__ __forceinline__ u32 pos(size_t i4) const { return (u32)((4 * i4) % ns); }
};

static __global__ void __launch_bounds__(kSortThreads) k_part_hist(const u32* __restrict__ digits, size_t row_len, size_t chunk_len, u32 nchunks,
                                                          u32 np, int low_bits, u32* __restrict__ hist, RowReal rr) {
    __shared__ u32 cnt[kMaxParts];
    const u32 chunk = blockIdx.x % nchunks, row = blockIdx.x / nchunks;
    for (u32 i = threadIdx.x; i < np; i += kSortThreads) cnt[i] = 0u;
    __syncthreads();
    const size_t c0 = (size_t)chunk * chunk_len;                              // multiple of 4
    const size_t c1 = (c0 + chunk_len < row_len) ? c0 + chunk_len : row_len;  // row_len multiple of 4
    const uint4* row4 = reinterpret_cast<const uint4*>(digits + (size_t)row * row_len);
    const u32 n4 = rr.n4(row), step = (u32)((4u * kSortThreads) % rr.ns);
    u32 pos = rr.pos((c0 >> 2) + threadIdx.x);
    for (size_t i4 = (c0 >> 2) + threadIdx.x; i4 < (c1 >> 2); i4 += kSortThreads) {
        const bool real = pos < n4;
        pos += step;
        if (pos >= rr.ns) pos -= rr.ns;
        if (!real) continue;
        uint4 d4 = row4[i4];
        u32 dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (dd[k] != kSkip) atomicAdd(&cnt[(dd[k] & 0x7fffffffu) >> low_bits], 1u);
    }
    __syncthreads();
    for (u32 p = threadIdx.x; p < np; p += kSortThreads) hist[((size_t)row * np + p) * nchunks + chunk] = cnt[p];
}

// exclusive scan of n <= blockDim values held one per thread (blockDim a power of two <= 1024)
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* sh, u32* total) {
    const int tid = threadIdx.x, n = blockDim.x;
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < n; off <<= 1) {
        u32 t = (tid >= off) ? sh[tid - off] : 0u;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    const u32 incl = sh[tid];
    if (total) *total = sh[n - 1];
    __syncthreads();
    return incl - v;
}

static constexpr int kScanThreads = 1024;  // >= kMaxParts (k_part_bases scans one row's partitions in a block)
static __global__ void __launch_bounds__(kScanThreads) k_part_scan(u32* __restrict__ hist, u32 nchunks, u32* __restrict__ total) {
    __shared__ u32 sh[kScanThreads];
    u32* h = hist + (size_t)blockIdx.x * nchunks;  // blockIdx.x = row * np + p
    u32 carry = 0;
    for (u32 t0 = 0; t0 < nchunks; t0 += blockDim.x) {
        const u32 i = t0 + threadIdx.x;
        const u32 v = (i < nchunks) ? h[i] : 0u;
        u32 tile_total;
        const u32 ex = block_exclusive_scan(v, sh, &tile_total);
        if (i < nchunks) h[i] = carry + ex;
        carry += tile_total;
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry;
}
static __global__ void __launch_bounds__(kScanThreads) k_part_bases(const u32* __restrict__ total, u32 np, u32* __restrict__ base,
                                                           u32* __restrict__ rowtot) {
    __shared__ u32 sh[kScanThreads];
    const u32 row = blockIdx.x;
    const u32 v = (threadIdx.x < np) ? total[(size_t)row * np + threadIdx.x] : 0u;
    u32 all;
    const u32 ex = block_exclusive_scan(v, sh, &all);
    if (threadIdx.x < np) base[(size_t)row * np + threadIdx.x] = ex;
    if (threadIdx.x == 0) rowtot[row] = all;
}

static __global__ void __launch_bounds__(kSortThreads) k_part_scatter(const u32* __restrict__ digits, size_t row_len, size_t chunk_len, u32 nchunks,
                                                             u32 np, int low_bits, const u32* __restrict__ hist,
                                                             const u32* __restrict__ base, int idx_bits, u32* __restrict__ part_idx,
                                                             unsigned short* __restrict__ part_low, RowReal rr) {
    __shared__ u32 cur[kMaxParts];
    const u32 chunk = blockIdx.x % nchunks, row = blockIdx.x / nchunks;
    for (u32 p = threadIdx.x; p < np; p += kSortThreads)
        cur[p] = base[(size_t)row * np + p] + hist[((size_t)row * np + p) * nchunks + chunk];
    __syncthreads();
    const size_t c0 = (size_t)chunk * chunk_len;
    const size_t c1 = (c0 + chunk_len < row_len) ? c0 + chunk_len : row_len;
    const uint4* row4 = reinterpret_cast<const uint4*>(digits + (size_t)row * row_len);
    u32* oi = part_idx + (size_t)row * row_len;
    unsigned short* ol = part_low + (size_t)row * row_len;
    const u32 low_mask = (1u << low_bits) - 1u;
    const u32 n4 = rr.n4(row), step = (u32)((4u * kSortThreads) % rr.ns);
    u32 seg = rr.pos((c0 >> 2) + threadIdx.x);
    for (size_t i4 = (c0 >> 2) + threadIdx.x; i4 < (c1 >> 2); i4 += kSortThreads) {
        const bool real = seg < n4;
        seg += step;
        if (seg >= rr.ns) seg -= rr.ns;
        if (!real) continue;
        uint4 d4 = row4[i4];
        u32 dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 d = dd[k];
            if (d != kSkip) {
                const u32 b = d & 0x7fffffffu;
                const u32 pos = atomicAdd(&cur[b >> low_bits], 1u);
                if (idx_bits) {  // index, bucket-in-partition and sign fit one word: a single store stream
                    oi[pos] = (u32)(4 * i4 + k) | ((b & low_mask) << idx_bits) | (d & 0x80000000u);
                } else {
                    oi[pos] = (u32)(4 * i4 + k) | (d & 0x80000000u);
                    ol[pos] = (unsigned short)(b & low_mask);
                }
            }
        }
    }
}

// the same scatter with the chunk (<= kStageChunk entries) first sorted by partition in LDS: every
// partition's run then leaves with consecutive-address stores (a run is ~chunk/np entries) instead of one
// 4-byte store per lane into np different streams.  Used for long rows, where the open store streams of
// all resident workgroups no longer fit the L2 (measured at 2^24: 4.0 ms -> see DESIGN.md).
static constexpr u32 kStageChunk = 16384;
static __global__ void __launch_bounds__(kSortThreads) k_part_scatter_staged(const u32* __restrict__ digits, size_t row_len, size_t chunk_len,
                                                                    u32 nchunks, u32 np, int low_bits, const u32* __restrict__ hist,
                                                                    const u32* __restrict__ base, int idx_bits,
                                                                    u32* __restrict__ part_idx, unsigned short* __restrict__ part_low, RowReal rr) {
    extern __shared__ u32 sm[];
    u32* cnt = sm;                    // [kMaxParts] entries of this chunk per partition, then fill cursors
    u32* loc = cnt + kMaxParts;       // [kMaxParts] start of the partition's run inside the stage
    u32* gcur = loc + kMaxParts;      // [kMaxParts] destination of that run in the row
    u32* sh = gcur + kMaxParts;       // [kSortThreads] scan scratch
    u32* st_idx = sh + kSortThreads;  // [kStageChunk]
    u32* st_meta = st_idx + kStageChunk;  // [kStageChunk] (partition << 16) | bucket-in-partition
    const u32 chunk = blockIdx.x % nchunks, row = blockIdx.x / nchunks;
    for (u32 p = threadIdx.x; p < kMaxParts; p += kSortThreads) cnt[p] = 0u;
    __syncthreads();
    const size_t c0 = (size_t)chunk * chunk_len;
    const size_t c1 = (c0 + chunk_len < row_len) ? c0 + chunk_len : row_len;
    const uint4* row4 = reinterpret_cast<const uint4*>(digits + (size_t)row * row_len);
    const u32 n4 = rr.n4(row), step = (u32)((4u * kSortThreads) % rr.ns);
    u32 seg = rr.pos((c0 >> 2) + threadIdx.x);
    for (size_t i4 = (c0 >> 2) + threadIdx.x; i4 < (c1 >> 2); i4 += kSortThreads) {
        const bool real = seg < n4;
        seg += step;
        if (seg >= rr.ns) seg -= rr.ns;
        if (!real) continue;
        uint4 d4 = row4[i4];
        u32 dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (dd[k] != kSkip) atomicAdd(&cnt[(dd[k] & 0x7fffffffu) >> low_bits], 1u);
    }
    __syncthreads();
    {
        const u32 v = cnt[threadIdx.x];  // kSortThreads == kMaxParts: one partition per thread
        u32 total;
        const u32 ex = block_exclusive_scan(v, sh, &total);
        loc[threadIdx.x] = ex;
        cnt[threadIdx.x] = ex;  // fill cursor
        if (threadIdx.x < np) gcur[threadIdx.x] = base[(size_t)row * np + threadIdx.x] + hist[((size_t)row * np + threadIdx.x) * nchunks + chunk];
        if (threadIdx.x == 0) sh[0] = total;
    }
    __syncthreads();
    const u32 staged = sh[0];
    const u32 low_mask = (1u << low_bits) - 1u;
    seg = rr.pos((c0 >> 2) + threadIdx.x);
    for (size_t i4 = (c0 >> 2) + threadIdx.x; i4 < (c1 >> 2); i4 += kSortThreads) {
        const bool real = seg < n4;
        seg += step;
        if (seg >= rr.ns) seg -= rr.ns;
        if (!real) continue;
        uint4 d4 = row4[i4];
        u32 dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 d = dd[k];
            if (d != kSkip) {
                const u32 b = d & 0x7fffffffu, p = b >> low_bits;
                const u32 pos = atomicAdd(&cnt[p], 1u);
                st_idx[pos] = (u32)(4 * i4 + k) | (d & 0x80000000u) | (idx_bits ? ((b & low_mask) << idx_bits) : 0u);
                st_meta[pos] = (p << 16) | (b & low_mask);
            }
        }
    }
    __syncthreads();
    u32* oi = part_idx + (size_t)row * row_len;
    unsigned short* ol = part_low + (size_t)row * row_len;
    for (u32 j = threadIdx.x; j < staged; j += kSortThreads) {
        const u32 meta = st_meta[j], p = meta >> 16;
        const u32 dst = gcur[p] + (j - loc[p]);
        oi[dst] = st_idx[j];
        if (!idx_bits) ol[dst] = (unsigned short)(meta & 0xffffu);
    }
}

// launched with 1024 threads for long partitions, 256 for short ones (fewer barrier steps in the scan)
static __global__ void __launch_bounds__(kSortThreads) k_part_sort(const u32* __restrict__ part_idx, const unsigned short* __restrict__ part_low,
                                                              size_t row_len, u32 np, int low_bits, int idx_bits, size_t nb,
                                                              const u32* __restrict__ base, const u32* __restrict__ rowtot,
                                                              uint2* __restrict__ oc, u32 T, size_t tiles_per_w, u32* __restrict__ tile_b,
                                                              u32* __restrict__ sorted) {
    __shared__ u32 cnt[kMaxLow];
    __shared__ u32 sh[kSortThreads];
    const u32 nthr = blockDim.x;
    const u32 p = blockIdx.x % np, row = blockIdx.x / np;
    const u32 nlow = 1u << low_bits, low_mask = nlow - 1u;
    const u32 keep = idx_bits ? (((1u << idx_bits) - 1u) | 0x80000000u) : 0xffffffffu;  // entry bits that survive (index | sign)
    const u32 s = base[(size_t)row * np + p];
    const u32 e = (p + 1 < np) ? base[(size_t)row * np + p + 1] : rowtot[row];
    const u32* pi = part_idx + (size_t)row * row_len;
    const unsigned short* pl = part_low + (size_t)row * row_len;
    for (u32 i = threadIdx.x; i < nlow; i += nthr) cnt[i] = 0u;
    __syncthreads();
    for (u32 i = s + threadIdx.x; i < e; i += nthr) {
        const u32 low = idx_bits ? ((pi[i] >> idx_bits) & low_mask) : (u32)pl[i];
        atomicAdd(&cnt[low], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0..nlow): each thread owns `per` consecutive counters
    const u32 per = (nlow + nthr - 1) / nthr;
    const u32 lo = threadIdx.x * per < nlow ? threadIdx.x * per : nlow, hi = (lo + per < nlow) ? lo + per : nlow;
    u32 sum = 0;
    for (u32 i = lo; i < hi; i++) sum += cnt[i];
    u32 run = s + block_exclusive_scan(sum, sh, nullptr);
    // (offset, count) of every bucket as one 8-byte record, and for every tile of the accumulation the bucket
    // its first entry falls into (tile t starts at entry t * T): the accumulation starts without a search
    uint2* oc_out = oc + (size_t)row * nb + (size_t)p * nlow;
    u32* tile_out = tile_b + (size_t)row * tiles_per_w;
    for (u32 i = lo; i < hi; i++) {
        const u32 v = cnt[i];
        oc_out[i] = make_uint2(run, v);
        cnt[i] = run;  // becomes the cursor of bucket i
        if (v)
            for (u32 t = (run + T - 1) / T; t <= (run + v - 1) / T; t++) tile_out[t] = p * nlow + i;
        run += v;
    }
    __syncthreads();
    u32* out = sorted + (size_t)row * row_len;
    for (u32 i = s + threadIdx.x; i < e; i += nthr) {
        const u32 v = pi[i];
        const u32 low = idx_bits ? ((v >> idx_bits) & low_mask) : (u32)pl[i];
        const u32 pos = atomicAdd(&cnt[low], 1u);
        out[pos] = v & keep;
    }
}


// ---------------------------------------------------------------------------------------
// Round 6: the sort of long WINDOW-TABLE rows without the digits array, and a level 2 that stores runs.
//
// Measured on the round-5 code at 2^24 points (13 windows of 20 bits: a row of 218 M entries; profiles/r06b_trace_2p24.txt):
// k_digits 0.26 + k_part_hist 0.26 + k_part_scan 0.13 + k_part_scatter_staged 1.07 + k_part_sort 2.91 = 4.63 ms, 9 GB of counter
// traffic against 1.4 GB compulsory (32 B per scalar in, 4 B per entry out).  Two causes:
//   * the digits took a round trip through HBM (52 B written per scalar, read twice) although a scalar is the most compact
//     form of its own digits: k_tab_hist / k_tab_scatter below recompute them from the 32-byte scalar (read twice, kept in
//     registers between the counting and the placing phase of the scatter);
//   * level 2 placed every entry with its own 4-byte store at a random position of its partition's 0.85 MB output range: 64
//     cache lines per wave store, two thirds of the phase.  the tiled level 2 (k_l2_*) sorts 16 Ki entries at a time by bucket in LDS and
//     writes every bucket's run (32+ entries) with consecutive lanes.
// The fused level 1 applies to classes that share one bucket set over all windows of a window table (`cl.shared`: the row is
// [W][ns], entry v = w * ns + i); other classes keep k_digits and the digit-reading kernels.
// ---------------------------------------------------------------------------------------
static constexpr int kTabW = 22;  // most windows of a fused class (12-bit tables: 22 windows)

// all digits of one canonical scalar over the W windows of the layout (W <= kTabW), in registers:
// d[w] = kSkip, or sign bit | (|digit| - 1) -- the same values k_digits stores.
// C > 0: the 256-bit layout of a C-bit window table (msm_layout(C, kFullBits)) as compile-time constants -- every digit is a
// bit-field at a known position of two known limbs (2-3 instructions instead of shifting the whole scalar: k_tab_hist at 2^24
// was bound by its ~700 instructions per scalar, not by the 32 bytes it reads).  C = 0: any layout, at run time.
// one window of the compile-time layout (recursion over w: every limb index and shift is a constant expression)
template <int C, int w>
__device__ __forceinline__ void recode_fixed(const u32 (&s)[8], u32 carry, u32 (&d)[kTabW]) {
    constexpr int W = (kFullBits + C - 1) / C, base = kFullBits / W, rem = kFullBits % W;
    static_assert(W <= kTabW && base + 1 <= 31, "layout");
    if constexpr (w < W) {
        constexpr int cw = base + (w < rem ? 1 : 0), off = w * base + (w < rem ? w : rem);
        constexpr int limb = off >> 5, sh = off & 31;
        constexpr u32 mask = (1u << cw) - 1u, half = 1u << (cw - 1);
        u32 v;
        if constexpr (sh + cw <= 32) v = (s[limb] >> sh) & mask;
        else v = ((s[limb] >> sh) | (s[limb + 1] << (32 - sh))) & mask;  // (off + cw <= 256: limb + 1 <= 7)
        v += carry;
        u32 sg = 0;
        if (v > half) {
            v = (1u << cw) - v;
            carry = 1;
            sg = 0x80000000u;
        } else {
            carry = 0;
        }
        d[w] = (v == 0) ? kSkip : (sg | (v - 1));
        recode_fixed<C, w + 1>(s, carry, d);
    } else if constexpr (w < kTabW) {
        d[w] = kSkip;
        recode_fixed<C, w + 1>(s, carry, d);
    }
}
template <int C>
__device__ __forceinline__ void recode_all(u32 (&s)[8], const WinLayout& L, u32 (&d)[kTabW]) {
    if constexpr (C > 0) {
        recode_fixed<C, 0>(s, 0u, d);
    } else {
        u32 carry = 0;
#pragma unroll
        for (int w = 0; w < kTabW; w++) {
            if (w < L.W) {
                const int cw = L.width(w);
                const u32 mask = (1u << cw) - 1u, half = 1u << (cw - 1);
                u32 v = (s[0] & mask) + carry;
#pragma unroll
                for (int k = 0; k < 7; k++) s[k] = (s[k] >> cw) | (s[k + 1] << (32 - cw));
                s[7] >>= cw;
                u32 sg = 0;
                if (v > half) {
                    v = (1u << cw) - v;
                    carry = 1;
                    sg = 0x80000000u;
                } else {
                    carry = 0;
                }
                d[w] = (v == 0) ? kSkip : (sg | (v - 1));
            } else {
                d[w] = kSkip;
            }
        }
    }
}

// 64-lane inclusive scan without LDS
__device__ __forceinline__ u32 wave_inclusive_scan(u32 v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}
// exclusive scan over the block (blockDim a multiple of 64, <= 1024) in three barriers; sh: >= 160 words
__device__ __forceinline__ u32 block_exclusive_scan_fast(u32 v, u32* sh, u32* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u32 incl = wave_inclusive_scan(v);
    if (lane == 63) sh[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        const u32 t = lane < nw ? sh[lane] : 0u;
        const u32 ti = wave_inclusive_scan(t);
        sh[64 + lane] = ti - t;
        if (lane == 63) sh[128] = ti;
    }
    __syncthreads();
    const u32 res = incl - v + sh[64 + wid];
    if (total) *total = sh[128];
    __syncthreads();
    return res;
}

// Workgroup b of a launch runs on XCD b mod 8, and every XCD has its own L2.  The scatters below write runs that CONTINUE the
// run of the previous chunk / tile (chunk c + 1's entries of partition p land right behind chunk c's): with consecutive chunks on
// different XCDs every 128-byte line at a seam is written partially from two L2s.  This map gives XCD x the x-th contiguous
// eighth of the chunks, in increasing order: seams close inside one L2.  `per` = ceil(count / 8); the launch has 8 * per
// workgroups per row, the ones that fall off the end exit.
__device__ __forceinline__ u32 xcd_contiguous(u32 b, u32 per) { return (b & 7u) * per + (b >> 3); }

// level 1, histogram: block (chunk, row) reads `chunk_sc` scalars of item `row` and counts their W digits per partition
template <int C>
static __global__ void __launch_bounds__(kSortThreads) k_tab_hist(const ItemDesc* __restrict__ items, WinLayout L, u32 chunk_sc, u32 nchunks, u32 np,
                                                         int low_bits, u32* __restrict__ hist) {
    __shared__ u32 cnt[kMaxParts];
    const u32 chunk = blockIdx.x % nchunks, row = blockIdx.x / nchunks;
    const ItemDesc it = items[row];
    for (u32 i = threadIdx.x; i < np; i += kSortThreads) cnt[i] = 0u;
    __syncthreads();
    const u32 i0 = chunk * chunk_sc;
    Fr sc[2];  // chunk_sc <= 2 * kSortThreads: both loads go out before the arithmetic
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const u32 i = i0 + q * kSortThreads + threadIdx.x;
        if ((u32)q * kSortThreads + threadIdx.x < chunk_sc && i < it.n) sc[q] = fr_load(it.scalars, i);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const u32 i = i0 + q * kSortThreads + threadIdx.x;
        if ((u32)q * kSortThreads + threadIdx.x >= chunk_sc || i >= it.n) break;
        Fr s = fp_from_mont<FrCfg>(sc[q]);
        u32 d[kTabW];
        recode_all<C>(s.l, L, d);
#pragma unroll
        for (int w = 0; w < kTabW; w++)
            if (d[w] != kSkip) atomicAdd(&cnt[(d[w] & 0x7fffffffu) >> low_bits], 1u);
    }
    __syncthreads();
    for (u32 p = threadIdx.x; p < np; p += kSortThreads) hist[((size_t)row * np + p) * nchunks + chunk] = cnt[p];
}

// level 1, scatter: the same chunk again.  How many of its entries fall into partition p is already in the scanned histogram
// (hist[p][chunk + 1] - hist[p][chunk]): no second counting pass over LDS atomics -- those, not the bytes, bound these kernels
// (~2 lane-atomics per clock and CU measured: 218 M entries cost ~0.2 ms per pass at 2^24).  The chunk's entries are grouped by
// partition in LDS (one word each: sign | window | scalar-in-chunk | bucket-in-partition) and every partition's run leaves with
// consecutive lanes (a half wave per run).  SPT = scalars per thread.
template <int SPT, int C>
static __global__ void __launch_bounds__(kSortThreads) k_tab_scatter(const ItemDesc* __restrict__ items, u32 ns, WinLayout L, u32 nchunks, u32 np,
                                                            int low_bits, const u32* __restrict__ hist, const u32* __restrict__ ptotal,
                                                            const u32* __restrict__ base, int idx_bits, size_t row_len, u32* __restrict__ part_idx,
                                                            unsigned short* __restrict__ part_low) {
    extern __shared__ u32 sm[];
    u32* cnt = sm;                    // [kMaxParts] fill cursors
    u32* loc = cnt + kMaxParts;       // [kMaxParts + 1] start of the partition's run inside the stage
    u32* gcur = loc + kMaxParts + 1;  // [kMaxParts] destination of that run in the row
    u32* sh = gcur + kMaxParts;       // [160] scan scratch
    u32* stage = sh + 160;            // [SPT * kSortThreads * W]
    const u32 per8 = (nchunks + 7) / 8, row = blockIdx.x / (8 * per8);
    const u32 chunk = xcd_contiguous(blockIdx.x % (8 * per8), per8);
    if (chunk >= nchunks) return;
    const ItemDesc it = items[row];
    const u32 i0 = chunk * (SPT * kSortThreads);
    if (i0 >= it.n) return;  // (a shorter item of the class: nothing in this chunk)
    // the scalars' loads go out first
    Fr sc[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const u32 i = i0 + k * kSortThreads + threadIdx.x;
        if (i < it.n) sc[k] = fr_load(it.scalars, i);
    }
    {
        u32 v = 0, h0 = 0;  // kSortThreads == kMaxParts: one partition per thread
        if (threadIdx.x < np) {
            const u32* h = hist + ((size_t)row * np + threadIdx.x) * nchunks;
            h0 = h[chunk];
            v = ((chunk + 1 < nchunks) ? h[chunk + 1] : ptotal[(size_t)row * np + threadIdx.x]) - h0;
            gcur[threadIdx.x] = base[(size_t)row * np + threadIdx.x] + h0;
        }
        u32 total;
        const u32 ex = block_exclusive_scan_fast(v, sh, &total);
        loc[threadIdx.x] = ex;
        cnt[threadIdx.x] = ex;
        if (threadIdx.x == 0) loc[kMaxParts] = total;
    }
    __syncthreads();
    const u32 low_mask = (1u << low_bits) - 1u;
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const u32 i = i0 + k * kSortThreads + threadIdx.x;
        if (i >= it.n) continue;
        Fr s = fp_from_mont<FrCfg>(sc[k]);
        u32 d[kTabW];
        recode_all<C>(s.l, L, d);
#pragma unroll
        for (int w = 0; w < kTabW; w++) {
            const u32 dd = d[w];
            if (dd != kSkip) {
                const u32 b = dd & 0x7fffffffu;
                const u32 pos = atomicAdd(&cnt[b >> low_bits], 1u);
                stage[pos] = (dd & 0x80000000u) | ((u32)w << 26) | ((u32)(k * kSortThreads + threadIdx.x) << 15) | (b & low_mask);  // low_bits <= 11 < 15, SPT * 1024 <= 2^11
            }
        }
    }
    __syncthreads();
    u32* oi = part_idx + (size_t)row * row_len;
    unsigned short* ol = part_low + (size_t)row * row_len;
    const u32 hw = threadIdx.x >> 5, l32 = threadIdx.x & 31;
    for (u32 p = hw; p < np; p += kSortThreads / 32) {
        const u32 a = loc[p], n = loc[p + 1] - a, g = gcur[p];
        for (u32 k = l32; k < n; k += 32) {
            const u32 wd = stage[a + k];
            const u32 v = ((wd >> 26) & 31u) * ns + i0 + ((wd >> 15) & 2047u);
            const u32 low = wd & 0x7fffu;
            if (idx_bits) {
                oi[g + k] = v | (low << idx_bits) | (wd & 0x80000000u);
            } else {
                oi[g + k] = v | (wd & 0x80000000u);
                ol[g + k] = (unsigned short)low;
            }
        }
    }
}

// level 2 for long partitions, three launches with a workgroup per TILE of kL2Tile entries (a first form with one workgroup per
// partition walking its tiles in sequence took 1.48 ms at 2^24: 1 024 workgroups, each a chain of 13 tiles, four rounds on the
// chip).  A partition of m entries is ceil(m / kL2Tile) tiles; tile j of the row belongs to partition p with tstart[p] <= j.
//   k_l2_tiles    block (row): tstart[row][0 .. np] = exclusive scan of the partitions' tile counts
//   k_l2_hist     block (tile, row): bucket histogram of the tile -> h2[row][tile][nlow]
//   k_l2_scan     block (p, row): per bucket the exclusive scan over the partition's tiles (h2 in place), the bucket counts,
//                 their scan -> (offset, count) records, first bucket of every accumulation tile (as k_part_sort)
//   k_l2_scatter  block (tile, row): entries in registers, grouped by bucket in LDS, every bucket's run stored by a half wave at
//                 oc[bucket].offset + h2[tile][bucket]
// Same outputs as k_part_sort (the order inside a bucket differs; the additions are exact).
static constexpr u32 kL2Per = 16, kL2Tile = kL2Per * kSortThreads;
static __global__ void __launch_bounds__(kScanThreads) k_l2_tiles(const u32* __restrict__ base, const u32* __restrict__ rowtot, u32 np, u32* __restrict__ tstart) {
    __shared__ u32 sh[kScanThreads];
    const u32 row = blockIdx.x, p = threadIdx.x;
    u32 v = 0;
    if (p < np) {
        const u32 s = base[(size_t)row * np + p], e = (p + 1 < np) ? base[(size_t)row * np + p + 1] : rowtot[row];
        v = (e - s + kL2Tile - 1) / kL2Tile;
    }
    u32 all;
    const u32 ex = block_exclusive_scan(v, sh, &all);
    if (p < np) tstart[(size_t)row * (np + 1) + p] = ex;
    if (p == 0) tstart[(size_t)row * (np + 1) + np] = all;
}
// tile j of `row` -> its partition and entry range; false: no such tile
__device__ __forceinline__ bool l2_tile_range(const u32* __restrict__ tstart, const u32* __restrict__ base, const u32* __restrict__ rowtot, u32 np, u32 row, u32 j,
                                              u32& p, u32& e0, u32& e1) {
    const u32* ts = tstart + (size_t)row * (np + 1);
    if (j >= ts[np]) return false;
    u32 lo = 0, hi = np;  // largest p with ts[p] <= j (partitions without tiles share their successor's start)
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (ts[mid] <= j) lo = mid;
        else hi = mid;
    }
    p = lo;
    const u32 s = base[(size_t)row * np + p], e = (p + 1 < np) ? base[(size_t)row * np + p + 1] : rowtot[row];
    e0 = s + (j - ts[p]) * kL2Tile;
    e1 = (e0 + kL2Tile < e) ? e0 + kL2Tile : e;
    return true;
}
static __global__ void __launch_bounds__(kSortThreads) k_l2_hist(const u32* __restrict__ part_idx, const unsigned short* __restrict__ part_low, size_t row_len, u32 np,
                                                        int low_bits, int idx_bits, const u32* __restrict__ base, const u32* __restrict__ rowtot,
                                                        const u32* __restrict__ tstart, u32 ub, u32* __restrict__ h2) {
    __shared__ u32 cnt[kMaxLow];
    const u32 j = blockIdx.x % ub, row = blockIdx.x / ub;
    u32 p, e0, e1;
    if (!l2_tile_range(tstart, base, rowtot, np, row, j, p, e0, e1)) return;
    const u32 nlow = 1u << low_bits, low_mask = nlow - 1u;
    for (u32 i = threadIdx.x; i < nlow; i += kSortThreads) cnt[i] = 0u;
    __syncthreads();
    const u32* pi = part_idx + (size_t)row * row_len;
    const unsigned short* pl = part_low + (size_t)row * row_len;
    for (u32 i = e0 + threadIdx.x; i < e1; i += kSortThreads) atomicAdd(&cnt[idx_bits ? ((pi[i] >> idx_bits) & low_mask) : (u32)pl[i]], 1u);
    __syncthreads();
    u32* h = h2 + ((size_t)row * ub + j) * nlow;
    for (u32 i = threadIdx.x; i < nlow; i += kSortThreads) h[i] = cnt[i];
}
static __global__ void __launch_bounds__(kSortThreads) k_l2_scan(u32 np, int low_bits, size_t nb, const u32* __restrict__ base, const u32* __restrict__ tstart, u32 ub,
                                                        u32* __restrict__ h2, uint2* __restrict__ oc, u32 T, size_t tiles_per_w, u32* __restrict__ tile_b) {
    __shared__ u32 sh[160];
    const u32 p = blockIdx.x % np, row = blockIdx.x / np;
    const u32 nlow = 1u << low_bits;
    const u32* ts = tstart + (size_t)row * (np + 1);
    const u32 j0 = ts[p], j1 = ts[p + 1];
    const u32 per = (nlow + kSortThreads - 1) / kSortThreads;  // 1 or 2 (nlow <= kMaxLow)
    const u32 lo = threadIdx.x * per < nlow ? threadIdx.x * per : nlow, hi = (lo + per < nlow) ? lo + per : nlow;
    u32 c[2] = {0u, 0u};
    for (u32 j = j0; j < j1; j++) {
        u32* h = h2 + ((size_t)row * ub + j) * nlow;
        for (u32 i = lo; i < hi; i++) {
            const u32 v = h[i];
            h[i] = c[i - lo];
            c[i - lo] += v;
        }
    }
    u32 run = base[(size_t)row * np + p] + block_exclusive_scan_fast(c[0] + c[1], sh, nullptr);
    uint2* oc_out = oc + (size_t)row * nb + (size_t)p * nlow;
    u32* tile_out = tile_b + (size_t)row * tiles_per_w;
    for (u32 i = lo; i < hi; i++) {
        const u32 v = c[i - lo];
        oc_out[i] = make_uint2(run, v);
        if (v)
            for (u32 t = (run + T - 1) / T; t <= (run + v - 1) / T; t++) tile_out[t] = p * nlow + i;
        run += v;
    }
}
static __global__ void __launch_bounds__(kSortThreads) k_l2_scatter(const u32* __restrict__ part_idx, const unsigned short* __restrict__ part_low, size_t row_len, u32 np,
                                                           int low_bits, int idx_bits, size_t nb, const u32* __restrict__ base, const u32* __restrict__ rowtot,
                                                           const u32* __restrict__ tstart, u32 ub, const u32* __restrict__ h2, const uint2* __restrict__ oc,
                                                           u32* __restrict__ sorted) {
    extern __shared__ u32 sm[];
    const u32 nlow = 1u << low_bits, low_mask = nlow - 1u;
    u32* lc = sm;              // [nlow] count, then cursor
    u32* loc = lc + nlow;      // [nlow + 1] start of the bucket's run in the stage
    u32* sh = loc + nlow + 1;  // [160]
    u32* stage = sh + 160;     // [kL2Tile]
    const u32 per8 = (ub + 7) / 8, row = blockIdx.x / (8 * per8);
    const u32 j = xcd_contiguous(blockIdx.x % (8 * per8), per8);
    u32 p, e0, e1;
    if (j >= ub || !l2_tile_range(tstart, base, rowtot, np, row, j, p, e0, e1)) return;
    const u32 keep = idx_bits ? (((1u << idx_bits) - 1u) | 0x80000000u) : 0xffffffffu;
    const u32* pi = part_idx + (size_t)row * row_len;
    const unsigned short* pl = part_low + (size_t)row * row_len;
    const u32* h = h2 + ((size_t)row * ub + j) * nlow;
    const uint2* ocp = oc + (size_t)row * nb + (size_t)p * nlow;
    u32 ev[kL2Per], el[kL2Per];
#pragma unroll
    for (u32 k = 0; k < kL2Per; k++) {  // (the loads go out first)
        const u32 i = e0 + k * kSortThreads + threadIdx.x;
        el[k] = 0xffffffffu;
        if (i < e1) {
            ev[k] = pi[i];
            el[k] = idx_bits ? ((ev[k] >> idx_bits) & low_mask) : (u32)pl[i];
        }
    }
    {
        // the tile's bucket counts are the differences of the scanned tile histograms (k_l2_scan) -- no counting pass
        const bool last = tstart[(size_t)row * (np + 1) + p + 1] == j + 1;
        const u32 per = (nlow + kSortThreads - 1) / kSortThreads;
        const u32 lo = threadIdx.x * per < nlow ? threadIdx.x * per : nlow, hi = (lo + per < nlow) ? lo + per : nlow;
        u32 c[2] = {0u, 0u};
        for (u32 i = lo; i < hi; i++) c[i - lo] = (last ? ocp[i].y : h[nlow + i]) - h[i];
        u32 total;
        u32 run = block_exclusive_scan_fast(c[0] + c[1], sh, &total);
        for (u32 i = lo; i < hi; i++) {
            loc[i] = run;
            lc[i] = run;
            run += c[i - lo];
        }
        if (threadIdx.x == 0) loc[nlow] = total;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < kL2Per; k++)
        if (el[k] != 0xffffffffu) stage[atomicAdd(&lc[el[k]], 1u)] = ev[k] & keep;
    __syncthreads();
    u32* out = sorted + (size_t)row * row_len;
    const u32 hw = threadIdx.x >> 5, l32 = threadIdx.x & 31;
    for (u32 b = hw; b < nlow; b += kSortThreads / 32) {
        const u32 a = loc[b], n = loc[b + 1] - a;
        if (n == 0) continue;
        const u32 g = ocp[b].x + h[b];
        for (u32 k = l32; k < n; k += 32) out[g + k] = stage[a + k];
    }
}

// host dispatch of the fused level 1 over the table widths that occur on long rows (any other width: the run-time layout, C = 0)
#define ZK_TAB_WIDTHS(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
static int tab_width_known(int c) {
#define X(Cc) if (c == Cc) return Cc;
    ZK_TAB_WIDTHS(X)
#undef X
    return 0;
}
static void tab_launch_hist(int c, dim3 grid, hipStream_t st, const ItemDesc* items, WinLayout L, u32 chunk_sc, u32 nchunks, u32 np, int low_bits, u32* hist) {
    switch (c) {
#define X(Cc) case Cc: hipLaunchKernelGGL(k_tab_hist<Cc>, grid, dim3(kSortThreads), 0, st, items, L, chunk_sc, nchunks, np, low_bits, hist); break;
        ZK_TAB_WIDTHS(X)
#undef X
        default: hipLaunchKernelGGL(k_tab_hist<0>, grid, dim3(kSortThreads), 0, st, items, L, chunk_sc, nchunks, np, low_bits, hist);
    }
}
template <int SPT, int C>
static void tab_launch_scatter_one(dim3 grid, size_t lds, hipStream_t st, const ItemDesc* items, u32 ns, WinLayout L, u32 nchunks, u32 np, int low_bits, const u32* hist,
                                   const u32* ptotal, const u32* base, int idx_bits, size_t row_len, u32* part_idx, unsigned short* part_low) {
    hipFuncSetAttribute((const void*)k_tab_scatter<SPT, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_tab_scatter<SPT, C>), grid, dim3(kSortThreads), lds, st, items, ns, L, nchunks, np, low_bits, hist, ptotal, base, idx_bits, row_len, part_idx, part_low);
}
static void tab_launch_scatter(int c, int spt, dim3 grid, size_t lds, hipStream_t st, const ItemDesc* items, u32 ns, WinLayout L, u32 nchunks, u32 np, int low_bits,
                               const u32* hist, const u32* ptotal, const u32* base, int idx_bits, size_t row_len, u32* part_idx, unsigned short* part_low) {
#define ZK_ARGS grid, lds, st, items, ns, L, nchunks, np, low_bits, hist, ptotal, base, idx_bits, row_len, part_idx, part_low
    switch (c) {
#define X(Cc) case Cc: if (spt == 2) tab_launch_scatter_one<2, Cc>(ZK_ARGS); else tab_launch_scatter_one<1, Cc>(ZK_ARGS); break;
        ZK_TAB_WIDTHS(X)
#undef X
        default: if (spt == 2) tab_launch_scatter_one<2, 0>(ZK_ARGS); else tab_launch_scatter_one<1, 0>(ZK_ARGS);
    }
#undef ZK_ARGS
}

// ---------------------------------------------------------------------------------------
// 4. bucket accumulation over fixed-size TILES of the sorted array: every lane performs exactly
// T mixed additions regardless of how the scalars distribute over buckets (no lock-step waiting
// on the fullest bucket of a wave, no collapse on skewed inputs).  A lane walks T consecutive
// sorted entries; runs that are whole buckets are stored directly, the (at most two) runs cut by
// the tile boundary go to heads[] / tails[] and are stitched by k_fixup.
// ---------------------------------------------------------------------------------------
template <class Cv>
static __global__ void __launch_bounds__(kBlk) k_accum_tiles(const ItemDesc* __restrict__ items, int rows_per_item,
                                                    const u32* __restrict__ sorted, const uint2* __restrict__ oc_all,
                                                    const u32* __restrict__ tile_b, size_t ns, u32 nsi, int nsi_shift, size_t nb, u32 T,
                                                    size_t tiles_per_w, size_t total_tiles, void* __restrict__ buckets,
                                                    void* __restrict__ heads, void* __restrict__ tails, int idx_ahead) {
  // one tile per lane; with a capped grid (tuning knob msm_share: a pass that leaves workgroup slots to other kernels) a lane
  // takes every gridDim.x * kBlk-th tile
  for (size_t g = (size_t)blockIdx.x * kBlk + threadIdx.x; g < total_tiles; g += (size_t)gridDim.x * kBlk) {
    const size_t w = g / tiles_per_w, t = g % tiles_per_w;  // w = row
    const ItemDesc it = items[w / rows_per_item];
    const void* __restrict__ bases = it.bases;
    // entry value v = index inside the row; a row spans `copies` images of the point range:
    // copy = v / nsi, point = v % nsi  ->  base index copy * pstride + point
    auto pidx = [&](u32 v) -> size_t {
        v &= 0x7fffffffu;
        if (it.pstride == 0) return v;
        const u32 win = nsi_shift >= 0 ? (v >> nsi_shift) : v / nsi;  // (rows of 2^k points: a shift instead of a ~25-instruction division)
        return (size_t)win * it.pstride + (v - win * nsi);
    };
    const uint2* oc = oc_all + w * nb;
    const uint2 last = oc[nb - 1];
    const u32 nw = last.x + last.y;  // entries of this window (zero digits are skipped)
    const u32 e0 = (u32)t * T;
    if (e0 >= nw) continue;
    const u32 e1 = (e0 + T < nw) ? e0 + T : nw;
    u32 b = tile_b[g];  // the bucket holding entry e0 (written by k_part_sort)
    uint2 cur_b = oc[b];
    u32 bstart = cur_b.x, bend = cur_b.x + cur_b.y;
    u32 ps = e0;  // start of the current run
    const u32* run = sorted + w * ns;
    typename Cv::Xyzz acc;
    Cv::set_inf(acc);
    u32 v = run[e0];
    // the index of the entry after next is fetched one iteration early: the gather of the next point then leaves at the top of the
    // iteration instead of behind the round trip of its own index (two dependent loads in a row were a stall of the whole wave)
    u32 vn = (idx_ahead && e0 + 1 < e1) ? run[e0 + 1] : 0u;
    typename Cv::Aff p = Cv::aff_load_rec(bases, pidx(v), it.rec);
    for (u32 e = e0; e < e1; e++) {
        if (e == bend) {  // run finished: flush and move to the next non-empty bucket
            if (ps == bstart) Cv::store(buckets, w * nb + b, acc);  // whole bucket
            else Cv::store(heads, g, acc);                            // started before this tile
            do {
                b++;
                cur_b = oc[b];
                bstart = cur_b.x;
                bend = cur_b.x + cur_b.y;
            } while (bend == bstart);
            ps = e;
            Cv::set_inf(acc);
        }
        const bool neg = (v >> 31) != 0;
        typename Cv::Aff cur = p;
        if (e + 1 < e1) {  // prefetch the next point while this one is being added
            if (idx_ahead) {
                v = vn;
                if (e + 2 < e1) vn = run[e + 2];
            } else {
                v = run[e + 1];
            }
            p = Cv::aff_load_rec(bases, pidx(v), it.rec);
        }
        Cv::madd(acc, cur, neg);
    }
    // last run of the tile
    if (ps == bstart && e1 == bend) Cv::store(buckets, w * nb + b, acc);
    else if (ps == e0) Cv::store(heads, g, acc);  // single run covering the tile from its start
    else Cv::store(tails, g, acc);
  }
}

// which bucket rows belong to a window one bit narrower than the widest: row r of the launch is window
// w0 + r % rpi of its item; such a row only ever uses the lower half of its buckets (and of every row the
// reduction derives from it), so the upper halves are neither written nor read.
struct NarrowRows {
    int rpi, w0, from;
    __host__ __device__ bool narrow(size_t row) const { return w0 + (int)(row % (size_t)rpi) >= from; }
};

// stitch the runs cut by tile boundaries; also writes infinity for empty buckets.  Buckets that
// span more than kLongSpan tiles (skewed scalars: many equal digits) are queued for k_fixup_long.
static constexpr u32 kLongSpan = 24;
template <class Cv>
static __global__ void __launch_bounds__(kBlk) k_fixup(const uint2* __restrict__ oc, size_t nb, NarrowRows nr, u32 T,
                                              size_t tiles_per_w, size_t total, void* __restrict__ buckets,
                                              const void* __restrict__ heads, const void* __restrict__ tails,
                                              u32* __restrict__ long_count, u32* __restrict__ long_list) {
    const size_t g = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (g >= total) return;
    const size_t w = g / nb;
    if (nr.narrow(w) && (g % nb) >= nb / 2) return;  // never populated, never read by the reduction
    const uint2 ocg = oc[g];
    const u32 s = ocg.x, c = ocg.y;
    if (c == 0) {
        typename Cv::Xyzz z;
        Cv::set_inf(z);
        Cv::store(buckets, g, z);
        return;
    }
    const u32 e = s + c;
    const u32 t0 = s / T, t1 = (e - 1) / T;
    if (t0 == t1) return;  // the whole bucket sat inside one tile and was stored by k_accum_tiles
    if (t1 - t0 > kLongSpan) {
        long_list[atomicAdd(long_count, 1u)] = (u32)g;
        return;
    }
    const size_t base = w * tiles_per_w;
    typename Cv::Xyzz acc = (s == t0 * T) ? Cv::load(heads, base + t0) : Cv::load(tails, base + t0);
    for (u32 t = t0 + 1; t <= t1; t++) acc = Cv::add(acc, Cv::load(heads, base + t));
    Cv::store(buckets, g, acc);
}

// the same with one bucket per QUAD of lanes (small and mid-size MSMs: the chain of dependent additions
// is pure latency, a quad runs each in 4 multiplication rounds instead of 13)
template <class Cv>
static __global__ void __launch_bounds__(kBlk) k_fixup_quad(const uint2* __restrict__ oc, size_t nb, NarrowRows nr, u32 T,
                                                   size_t tiles_per_w, size_t total, void* __restrict__ buckets,
                                                   const void* __restrict__ heads, const void* __restrict__ tails,
                                                   u32* __restrict__ long_count, u32* __restrict__ long_list) {
    const size_t tq = (size_t)blockIdx.x * kBlk + threadIdx.x;
    const size_t g = tq >> 2;
    const int role = (int)(tq & 3);
    if (g >= total) return;
    const size_t w = g / nb;
    if (nr.narrow(w) && (g % nb) >= nb / 2) return;  // never populated, never read by the reduction
    const uint2 ocg = oc[g];
    const u32 s = ocg.x, c = ocg.y;
    if (c == 0) {
        Cv::quad_store_inf(buckets, g, role);
        return;
    }
    const u32 e = s + c;
    const u32 t0 = s / T, t1 = (e - 1) / T;
    if (t0 == t1) return;  // the whole bucket sat inside one tile and was stored by k_accum_tiles
    if (t1 - t0 > kLongSpan) {
        if (role == 0) long_list[atomicAdd(long_count, 1u)] = (u32)g;
        return;
    }
    const size_t base = w * tiles_per_w;
    typename Cv::Xyzz acc = (s == t0 * T) ? Cv::load(heads, base + t0) : Cv::load(tails, base + t0);
    for (u32 t = t0 + 1; t <= t1; t++) acc = Cv::acc_quad(acc, heads, base + t, role);
    Cv::quad_store(buckets, g, role, acc);
}

// one workgroup per long bucket: strided partial sums per lane, then an LDS tree
template <class Cv>
static __global__ void __launch_bounds__(kBlk) k_fixup_long(const uint2* __restrict__ oc, size_t nb, u32 T,
                                                   size_t tiles_per_w, void* __restrict__ buckets, const void* __restrict__ heads,
                                                   const void* __restrict__ tails, const u32* __restrict__ long_count,
                                                   const u32* __restrict__ long_list) {
    extern __shared__ uint4 red[];  // blockDim.x XYZZ points
    const u32 nlong = *long_count;
    for (u32 i = blockIdx.x; i < nlong; i += gridDim.x) {
        const size_t g = long_list[i];
        const size_t w = g / nb;
        const uint2 ocg = oc[g];
        const u32 s = ocg.x, e = s + ocg.y;
        const u32 t0 = s / T, t1 = (e - 1) / T;
        const size_t base = w * tiles_per_w;
        typename Cv::Xyzz acc;
        Cv::set_inf(acc);
        for (u32 k = threadIdx.x; k <= t1 - t0; k += blockDim.x) {
            typename Cv::Xyzz piece = (k == 0 && s != t0 * T) ? Cv::load(tails, base + t0) : Cv::load(heads, base + t0 + k);
            acc = Cv::add(acc, piece);
        }
        Cv::store(red, threadIdx.x, acc);
        __syncthreads();
        for (int stride = (int)blockDim.x / 2; stride > 0; stride >>= 1) {
            if ((int)threadIdx.x < stride) {
                typename Cv::Xyzz a = Cv::load(red, threadIdx.x), b2 = Cv::load(red, threadIdx.x + stride);
                Cv::store(red, threadIdx.x, Cv::add(a, b2));
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) Cv::store(buckets, g, Cv::load(red, 0));
        __syncthreads();
    }
}

// one bit-plane pass of the bucket reduction.  in: [W][rows][len], out: [W][rows+1][len/2]
template <class Cv>
static __global__ void __launch_bounds__(kBlk) k_halve(const void* __restrict__ in, void* __restrict__ out, int W, int rows, size_t len, NarrowRows nr) {
    const size_t half = len >> 1;
    const size_t per_w = (size_t)(rows + 1) * half;
    const size_t t = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (t >= per_w * W) return;
    const size_t w = t / per_w, rem = t % per_w;
    const int r = (int)(rem / half);
    const size_t j = rem % half;
    const size_t eff = nr.narrow(w) ? half : len;  // elements of every row of this window that can be non-trivial
    if (2 * j >= eff) return;
    const size_t in_w = w * (size_t)rows * len;
    const size_t src_row = (r < rows - 1) ? (size_t)r : (size_t)(rows - 1);  // rows-1 = the L row
    typename Cv::Xyzz b;
    if (2 * j + 1 < eff) b = Cv::load(in, in_w + src_row * len + 2 * j + 1);
    else Cv::set_inf(b);
    typename Cv::Xyzz res;
    if (r == rows - 1) {
        res = b;  // odd elements of L become the new plane row
    } else {
        typename Cv::Xyzz a = Cv::load(in, in_w + src_row * len + 2 * j);
        res = Cv::add(a, b);
    }
    Cv::store(out, t, res);
}

// the same pass with one addition per QUAD of lanes (curve30.cuh: xyzz30_add_quad; curve30_g2.cuh: xyzz2_add_quad): for the
// late passes, which are a single dependent addition of pure latency, this cuts the chain from 13 multiplications to 4
template <class Cv>
static __global__ void __launch_bounds__(kBlk) k_halve_quad(const void* __restrict__ in, void* __restrict__ out, int W, int rows, size_t len,
                                                   NarrowRows nr) {
    const size_t tq = (size_t)blockIdx.x * kBlk + threadIdx.x;
    const size_t half = len >> 1;
    const size_t per_w = (size_t)(rows + 1) * half;
    const size_t t = tq >> 2;
    const int role = (int)(tq & 3);
    if (t >= per_w * W) return;  // whole quads leave together (kBlk is a multiple of 4)
    const size_t w = t / per_w, rem = t % per_w;
    const int r = (int)(rem / half);
    const size_t j = rem % half;
    const size_t eff = nr.narrow(w) ? half : len;
    if (2 * j >= eff) return;
    const size_t in_w = w * (size_t)rows * len;
    const size_t src_row = (r < rows - 1) ? (size_t)r : (size_t)(rows - 1);
    const size_t ib = in_w + src_row * len + 2 * j + 1;
    const bool b_real = 2 * j + 1 < eff;
    if (r == rows - 1) {  // odd elements of L become the new plane row: lane r copies coordinate r
        if (b_real) Cv::quad_copy(in, ib, out, t, role);
        else Cv::quad_store_inf(out, t, role);
        return;
    }
    if (!b_real) {  // a + infinity
        Cv::quad_copy(in, ib - 1, out, t, role);
        return;
    }
    Cv::add_quad(in, ib - 1, ib, out, t, role);
}

// last step on the device: the c reduced points of every window row (planes T_0..T_{c-2}, then T_all)
// are rewritten as Jacobian points in the REFERENCE Montgomery form (144 B), so the host chain starts
// without conversions.  With `pair` the planes are also combined two by two on the way,
// U_j = T_2j + 2*T_{2j+1} (+ T_all for j = 0), halving the host's additions at the price of a longer
// device chain (measured: a wash at 2^20).
// in: [rows][c] XYZZ, out: [rows][nout] Jacobian; nout = c, or max(1, c/2) when pairing.
template <class Cv>
static __global__ void __launch_bounds__(64) k_finish(const void* __restrict__ in, void* __restrict__ out, size_t rows, int c, int nout, int pair) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= rows * (size_t)nout) return;
    const size_t row = t / nout;
    const int j = (int)(t % nout);
    typename Cv::Xyzz acc;
    if (!pair) {
        acc = Cv::load(in, row * c + j);
    } else {
        const int planes = c - 1;  // T_0 .. T_{c-2}; index c-1 is T_all
        Cv::set_inf(acc);
        if (2 * j < planes) acc = Cv::load(in, row * c + 2 * j);
        if (j == 0) acc = Cv::add(acc, Cv::load(in, row * c + (c - 1)));
        if (2 * j + 1 < planes) acc = Cv::add(acc, Cv::dbl(Cv::load(in, row * c + 2 * j + 1)));
    }
    Cv::finish(acc, out, t);
}

// SRS precomputation: table[w][i] = 2^{bit_offset(w)} * P_i (affine records), one lane per point.
// With it every window's digit can use the SAME bucket set (the factor 2^{c w} is in the base).
template <class Cv>
static __global__ void __launch_bounds__(Cv::kEndo ? kBlk : 64) k_precompute(const void* __restrict__ bases, size_t n, size_t nsr, WinLayout L,
                                                                       void* __restrict__ table, u32 rec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const typename Cv::Aff p = Cv::aff_load(bases, i);
    typename Cv::Xyzz acc;
    Cv::set_inf(acc);
    Cv::madd(acc, p, false);
    for (int w = 0; w < L.W; w++) {
        if (w > 0)
            for (int k = 0; k < L.width(w - 1); k++) acc = Cv::dbl(acc);
        Cv::table_store(acc, table, (size_t)w * nsr + i, rec);
    }
}

#ifndef ZK_MSM_TU_G2
// G1 tables, round 6: k_precompute above normalises every copy of every point with an inversion of its own -- 14 Fermat inversions
// (~570 multiplications each) beside 255 doublings (~2 000): 80 % of a table's build time.  The two kernels below split the work:
// the chain kernel stores the XYZZ form of every copy of a CHUNK of points ([W][cn_pad], the blocked XYZZ layout), the
// normalisation runs Montgomery's trick per lane over 64 strided entries of that array (one inversion per 64 records) and writes
// the records where the table wants them.  Same canonical coordinates, hence the same table bytes.
static __global__ void __launch_bounds__(kBlk) k_table_chain(const void* __restrict__ bases, size_t c0, size_t cn, size_t cn_pad, WinLayout L,
                                                    void* __restrict__ xyzz) {
    const size_t j = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (j >= cn) return;
    const Aff30 p = aff30_load(bases, c0 + j);
    Xyzz30 acc;
    xyzz30_set_inf(acc);
    xyzz30_madd(acc, p, false);
    for (int w = 0; w < L.W; w++) {
        if (w > 0)
            for (int k = 0; k < L.width(w - 1); k++) acc = xyzz30_dbl(acc);
        xyzz30_store(xyzz, (size_t)w * cn_pad + j, acc);
    }
}
// lane t owns the entries {t + i T} of the [W][cn_pad] array; entry idx = w * cn_pad + j becomes record (w * nsr + c0 + j) of the table
static __global__ void __launch_bounds__(kBlk) k_table_normalise(const void* __restrict__ in, size_t total, size_t T, void* __restrict__ prefix, size_t c0, size_t cn,
                                                        size_t cn_pad, size_t nsr, void* __restrict__ table, u32 rec) {
    const size_t t = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (t >= T || t >= total) return;
    const size_t cnt = (total - t + T - 1) / T;
    Fq30 p = f30_one();
    for (size_t i = 0; i < cnt; i++) {
        const size_t idx = t + i * T;
        f30_store(prefix, idx * 48, p);  // product of the ZZZ before this entry
        const Fq30 zzz = f30_load_chunks(in, idx, 9);
        if (!f30_all_zero(zzz)) p = f30_mul(p, zzz);
    }
    Fq30 inv = f30_inv(p);
    for (size_t i = cnt; i-- > 0;) {
        const size_t idx = t + i * T;
        const Fq30 zzz = f30_load_chunks(in, idx, 9);
        Fq30 x = f30_zero(), y = f30_zero();
        if (!f30_all_zero(zzz)) {
            const Fq30 i3 = f30_mul(inv, f30_load(prefix, idx * 48));  // 1 / ZZZ
            inv = f30_mul(inv, zzz);
            const Fq30 iz = f30_mul(f30_load_chunks(in, idx, 6), i3);  // ZZ / ZZZ = 1 / Z
            x = f30_canon8(f30_mul(f30_load_chunks(in, idx, 0), f30_sqr(iz)));
            y = f30_canon8(f30_mul(f30_load_chunks(in, idx, 3), i3));
        }
        const size_t w = idx / cn_pad, j = idx - w * cn_pad;
        if (j < cn) {  // (the padding of a chunk belongs to no record)
            f30_store(table, (w * nsr + c0 + j) * rec, x);
            f30_store(table, (w * nsr + c0 + j) * rec + 48, y);
        }
    }
}
// -> ZK_OK, or an error after which the caller falls back to k_precompute (no memory for the transient arrays)
static int precompute_table_g1_batched(zk_ctx* ctx, const zk_srs* srs, const WinLayout& L, size_t nsr, void* d_table, u32 rec) {
    const size_t chunk = std::min<size_t>(srs->n, (size_t)1 << 20);
    const size_t pad = (chunk + 63) & ~(size_t)63, total_max = (size_t)L.W * pad;
    void *d_x = nullptr, *d_pref = nullptr;
    hipError_t e = device_alloc(ctx, &d_x, total_max * 192);
    if (e == hipSuccess) e = device_alloc(ctx, &d_pref, total_max * 48);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (d_x) hipFree(d_x);
        return ZK_ERR_OOM;
    }
    for (size_t c0 = 0; c0 < srs->n && e == hipSuccess; c0 += chunk) {
        const size_t cn = std::min(chunk, srs->n - c0), cn_pad = (cn + 63) & ~(size_t)63, total = (size_t)L.W * cn_pad;
        e = hipMemsetAsync(d_x, 0, total * 192, ctx->stream);  // padding entries: ZZZ = 0 = infinity
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_table_chain, dim3((unsigned)((cn + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)srs->d_bases, c0, cn, cn_pad, L, d_x);
        const size_t T = std::max<size_t>(1, (total + 63) / 64);
        hipLaunchKernelGGL(k_table_normalise, dim3((unsigned)((T + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)d_x, total, T, d_pref, c0, cn, cn_pad, nsr,
                           d_table, rec);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_x);
    hipFree(d_pref);
    return e == hipSuccess ? ZK_OK : hip_fail(ctx, e, "window table (batched normalisation)");
}
#endif

// SRS format conversion: reference Montgomery form (x * 2^384) <-> internal (x * 2^390), in place or copy
static __global__ void __launch_bounds__(kBlk) k_srs_convert(const void* __restrict__ in, void* __restrict__ out, size_t ncoord, int to_internal) {
    for (size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x; i < ncoord; i += (size_t)gridDim.x * kBlk) {
        Fq30 v = f30_load(in, i * 48);
        f30_store(out, i * 48, to_internal ? f30_from_ref(v) : f30_to_ref(v));
    }
}

// second half of the device SRS: phi(P_i) = (beta * x_i, y_i) (infinity stays infinity: beta * 0 = 0)
static __global__ void __launch_bounds__(kBlk) k_srs_endo(void* __restrict__ bases, size_t n) {
    Fq30 beta;
    {
        constexpr u32 t[13] = {0x1c907181u, 0x3cbde486u, 0x26574c3eu, 0x332475ecu, 0x1c3ebc1bu, 0x39ee6864u, 0x16ffa856u,
                               0x2c3499ffu, 0x0550bd16u, 0x14cbac30u, 0x17d18c86u, 0x215959f7u, 0x0009c6d4u};
#pragma unroll
        for (int i = 0; i < 13; i++) beta.l[i] = t[i];
    }
    for (size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlk) {
        Aff30 p = aff30_load(bases, i);
        f30_store(bases, (n + i) * 96, f30_canon8(f30_mul(p.x, beta)));
        f30_store(bases, (n + i) * 96 + 48, p.y);
    }
}

#ifndef ZK_MSM_TU_G2  // (G1 translation unit only)
// test hooks: field ops on reference-form Fq vectors through the production (unsaturated) arithmetic
static __global__ void __launch_bounds__(kBlk) k_dbg_fq(const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ out, size_t n,
                                               int op) {
    for (size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlk) {
        Fq30 x = f30_from_ref(f30_load(a, i * 48)), y = f30_from_ref(f30_load(b, i * 48));
        Fq30 r = (op == 0) ? f30_add(x, y) : (op == 1) ? f30_sub2(x, y) : (op == 3) ? f30_mul2add(x, y, y, y) : (a == b) ? f30_sqr(x) : f30_mul(x, y);  // same buffer twice: the squaring path; op 3: x*y + y*y under one reduction
        f30_store(out, i * 48, f30_to_ref(r));
    }
}

// test hook: XYZZ arithmetic on pairs of affine points
static __global__ void __launch_bounds__(kBlk) k_dbg_g1(const void* __restrict__ p, const void* __restrict__ q, void* __restrict__ out,
                                               size_t n, int mode) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    Aff30 a, b;  // reference-form inputs -> internal form
    a.x = f30_from_ref(f30_load(p, i * 96));
    a.y = f30_from_ref(f30_load(p, i * 96 + 48));
    b.x = f30_from_ref(f30_load(q, i * 96));
    b.y = f30_from_ref(f30_load(q, i * 96 + 48));
    Xyzz30 s;
    xyzz30_set_inf(s);
    xyzz30_madd(s, a, false);
    xyzz30_madd(s, b, mode == 3);  // p + q   (mode 3: p - q)
    Xyzz30 r = s;
    if (mode == 1) {  // (p+q) + p
        Xyzz30 pa;
        xyzz30_set_inf(pa);
        xyzz30_madd(pa, a, false);
        r = xyzz30_add(s, pa);
    } else if (mode == 2) {  // (p+q) + (p+q): doubling path of the full addition
        r = xyzz30_add(s, s);
    }
    xyzz30_store_flat(out, i, r);
}

#endif
#ifdef ZK_MSM_TU_G2  // (G2 translation unit only)
// test hook: the G2 formulas on pairs of affine points (reference form, 192 B); output through CvG2::finish (288 B Jacobian)
// mode 0: p + q   1: (p + q) + p   2: (p + q) + (p + q)   3: p - q   4: 2 (p + q) by the doubling   5: (p + q) - (p + q)
static __global__ void __launch_bounds__(64) k_dbg_g2(const void* __restrict__ p, const void* __restrict__ q, void* __restrict__ out, size_t n, int mode) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Aff2 a, b;
    a.x = Fq2x{f30_from_ref(f30_load(p, i * 192)), f30_from_ref(f30_load(p, i * 192 + 48))};
    a.y = Fq2x{f30_from_ref(f30_load(p, i * 192 + 96)), f30_from_ref(f30_load(p, i * 192 + 144))};
    b.x = Fq2x{f30_from_ref(f30_load(q, i * 192)), f30_from_ref(f30_load(q, i * 192 + 48))};
    b.y = Fq2x{f30_from_ref(f30_load(q, i * 192 + 96)), f30_from_ref(f30_load(q, i * 192 + 144))};
    Xyzz2 s;
    xyzz2_set_inf(s);
    xyzz2_madd(s, a, false);
    xyzz2_madd(s, b, mode == 3);
    Xyzz2 r = s;
    if (mode == 1) {
        Xyzz2 pa;
        xyzz2_set_inf(pa);
        xyzz2_madd(pa, a, false);
        r = xyzz2_add(s, pa);
    } else if (mode == 2) {
        r = xyzz2_add(s, s);
    } else if (mode == 4) {
        r = xyzz2_dbl(s);
    } else if (mode == 5) {
        Xyzz2 m = s;
        m.y = f2_sub4(f2_zero(), s.y);  // -(p + q): 4q - Y < 4q
        r = xyzz2_add(s, m);
    }
    CvG2::finish(r, out, i);
}

#endif
// ---------------------------------------------------------------------------------------
// device XYZZ (internal Montgomery form 2^390, coordinates < 8q) -> host Jacobian in the reference form:
// one host multiplication by 2^-6 per coordinate (K = 2^378 mod q = the 2^384-form of 2^-6)
static zkhost::Jac load_xyzz_host(const uint64_t* p) {
    zkhost::Fq X, Y, ZZ, ZZZ;
    std::memcpy(X.data(), p, 48);
    std::memcpy(Y.data(), p + 6, 48);
    std::memcpy(ZZ.data(), p + 12, 48);
    std::memcpy(ZZZ.data(), p + 18, 48);
    if (zkhost::is_zero(ZZ)) return zkhost::jac_inf();
    return zkhost::xyzz_to_jac(zkhost::mul(X, zkhost::K378()), zkhost::mul(Y, zkhost::K378()), zkhost::mul(ZZ, zkhost::K378()),
                               zkhost::mul(ZZZ, zkhost::K378()));
}

// host combine: per window row `nout` Jacobian points (k_finish) sit at their bit position -- plane k
// of window w at bit_offset(w) + k, T_all at bit_offset(w); with pairing U_j at bit_offset(w) + 2j --
// and result = sum_p 2^p * pos[p] is one doubling chain from the top position down.  The chain can be
// run in pieces: combine_windows() continues `acc` through the positions of windows [w0, w0 + wc),
// which must be the next lower ones (a class may hold only part of the windows, highest part first).
// (Splitting the chain over host threads was measured: thread start-up costs what the shorter chain saves.)
template <class HF>
static void combine_windows(zkhost::JacT<HF>& acc, const uint64_t* h, const WinLayout& L, int w0, int wc, int c, int nout, bool pair) {
    typedef zkhost::JacT<HF> J;
    constexpr size_t fw = zkhost::fe_words<HF>();
    const size_t p_lo = (size_t)L.bit_offset(w0);
    const size_t p_hi = (w0 + wc >= L.W) ? (size_t)L.bit_offset(L.W - 1) + L.width(L.W - 1) + c + 2 : (size_t)L.bit_offset(w0 + wc);
    std::vector<J> pos(p_hi - p_lo, zkhost::jac_inf_t<HF>());
    for (int w = 0; w < wc; w++) {
        for (int j = 0; j < nout; j++) {
            const uint64_t* src = h + ((size_t)w * nout + j) * 3 * fw;
            J pt;
            zkhost::get_fe(pt.x, src);
            zkhost::get_fe(pt.y, src + fw);
            zkhost::get_fe(pt.z, src + 2 * fw);
            if (zkhost::is_zero(pt.z)) continue;
            const size_t p = (size_t)L.bit_offset(w0 + w) + (pair ? 2 * (size_t)j : (j == c - 1 ? 0 : (size_t)j)) - p_lo;
            pos[p] = zkhost::jac_add(pos[p], pt);
        }
    }
    for (size_t p = pos.size(); p-- > 0;) {
        acc = zkhost::jac_dbl(acc);  // doubling the identity is free
        if (!zkhost::is_zero(pos[p].z)) acc = zkhost::jac_add(acc, pos[p]);
    }
}

// Host worker pool of a ctx (created on first use, joined by zk_ctx_destroy): the per-item doubling
// chains run here while the calling thread is still waiting for the device work of later classes.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::deque<std::function<void()>> q;
    size_t pending = 0;
    bool stop = false;
    explicit HostPool(unsigned n) {
        for (unsigned i = 0; i < n; i++)
            th.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return stop || !q.empty(); });
                        if (q.empty()) return;
                        job = std::move(q.front());
                        q.pop_front();
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    void submit(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(f));
            pending++;
        }
        cv.notify_one();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this] { return pending == 0; });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};
static HostPool* host_pool(zk_ctx* ctx) {
    if (!ctx->host_pool) ctx->host_pool = new HostPool(std::min(64u, std::max(2u, std::thread::hardware_concurrency() / 2)));
    return (HostPool*)ctx->host_pool;
}
#ifndef ZK_MSM_TU_G2  // (G1 translation unit only)
void msm_host_pool_destroy(zk_ctx* ctx) {
    delete (HostPool*)ctx->host_pool;
    ctx->host_pool = nullptr;
}

#endif
struct MsmClass {
    bool shared = false;  // precomputed-table mode: one bucket row per item spanning all windows
    int rpi = 0;          // bucket rows per item (the class's windows, or 1 when shared)
    size_t row_len = 0;   // entries per row: 2 * ns (points and their endomorphism images), or W * ns when shared
    int copies = 2;
    int c = 0;            // bits of the widest window: 2^(c-1) buckets per row, c reduced points per row
    int key_c = 0;        // the window bits asked for (class key)
    int size_key = 0;     // window-table classes: log2 of the item length (class key)
    int narrow_from = 0x7fffffff;  // windows with index >= this are one bit narrower
    int w0 = 0, wc = 0;   // the windows [w0, w0 + wc) of the layout this class works on
    int part = 0, nparts = 1;  // a big class is cut by windows into parts that run staggered (see below)
    int npair = 0;        // points per window row handed to the host (k_finish)
    bool pair = false;
    WinLayout L{};
    std::vector<size_t> idx;  // items of the batch in this class
    size_t ns = 0, nb = 0, rows = 0, total = 0, tiles_per_w = 0, total_tiles = 0, chunk_len = 0, cc_elems = 0;
    u32 T = 32, nchunks = 0, np = 1;  // sort: chunks per row, partitions per row
    int low_bits = 0;                 // log2(buckets per partition)
    int idx_bits = 0;                 // > 0: level-1 entries carry the bucket-in-partition above the index bits
    bool staged_scatter = false;      // level-1 scatter through an LDS-sorted chunk (long rows)
    bool fused_tab = false;           // level 1 straight from the scalars (k_tab_hist / k_tab_scatter): long window-table rows
    bool tiled_l2 = false;            // level 2 with LDS-staged runs (k_l2_*): long partitions
    u32 tab_spt = 2;                  // fused level 1: scalars per thread (chunk = tab_spt * 1024 scalars)
    int tab_c = 0;                    // fused level 1: the table width when a compile-time layout exists for it (else 0)
    u32 l2_ub = 0;                    // tiled level 2: upper bound of the tiles of a row (row_len / kL2Tile + np)
    size_t pinned_off = 0;  // byte offset of this class's results in the pinned staging area
    size_t off[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // byte offsets of this class inside the scratch arenas
};

// batches group their items into window classes: 5, 7, ..., 17 bits (rounded up), then 19.  Step 2 keeps the
// padding of a class (rows are as long as its largest item) below 4x; measured 5 % faster end to end than step 3
static int quantised_window(int c) {
    const int step = std::max(1, (int)tuning().msm_qstep);
    return c <= 5 ? 5 : (c > 17 ? 19 : std::min(17, 5 + step * ((c - 5 + step - 1) / step)));
}

// One batch in flight: what msm_enqueue leaves behind for msm_finish.  A blocking call (lane 0) works in the ctx's scratch
// arenas and pinned staging area; an asynchronous job (lanes 1, 2) owns its buffers -- taken from the ctx's block pool and
// handed back when the job is waited for -- because sumcheck calls and other jobs run while it is in flight.
struct MsmRun {
    int lane = 0;
    bool own_mem = false;
    size_t count = 0;
    std::vector<MsmItem> items;
    std::vector<MsmClass> classes;
    std::vector<size_t> empty_items;  // n = 0: the result is the identity
    void* buf[10] = {};
    char* hpin = nullptr;
    size_t pinned_bytes = 0;
    size_t accum_wg_cap = 0;  // workgroups of every k_accum_tiles launch of this pass (0: one lane per tile)
};

static void msm_release(zk_ctx* ctx, MsmRun& run) {
    if (!run.own_mem || run.lane <= 0) return;
    ctx->lanes[run.lane].busy = false;  // (the lane keeps its arenas for the next job)
    run.lane = -1;
}

static int msm_lane_prepare(zk_ctx* ctx, int lane) {
    zk_ctx::MsmLane& L = ctx->lanes[lane];
    if (lane == 0) L.main = ctx->stream;
    if (L.ready) return ZK_OK;
    ZK_HIP(ctx, hipStreamCreateWithFlags(&L.main, hipStreamNonBlocking));
    for (auto& s2 : L.aux) ZK_HIP(ctx, hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    ZK_HIP(ctx, hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming));
    for (auto& e : L.ev_join) ZK_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : L.ev_part) ZK_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    L.ready = true;
    return ZK_OK;
}
#ifndef ZK_MSM_TU_G2  // (G1 translation unit only)
// zk_arena_plan_import: an idle asynchronous lane's arenas grown to the given sizes
int msm_lanes_reserve(zk_ctx* ctx, int lane, const uint64_t* caps10, uint64_t pinned_cap) {
    if (lane < 1 || lane >= zk_ctx::kLanes) return fail(ctx, ZK_ERR_INVALID, "arena plan: lane out of range");
    zk_ctx::MsmLane& L = ctx->lanes[lane];
    if (L.busy) return fail(ctx, ZK_ERR_INVALID, "arena plan: an MSM job is in flight on lane %d", lane);
    bool any = pinned_cap != 0;
    for (int i = 0; i < 10; i++) any = any || caps10[i] != 0;
    if (!any) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    {
        const int rc = msm_lane_prepare(ctx, lane);
        if (rc) return rc;
    }
    for (int i = 0; i < 10; i++) {
        zk_ctx::Arena& a = L.mem[i];
        if (a.cap >= caps10[i]) continue;
        if (a.p) hipFree(a.p);
        a.p = nullptr, a.cap = 0;
        const hipError_t e = device_alloc(ctx, &a.p, (size_t)caps10[i]);
        if (e != hipSuccess) {
            a.p = nullptr;
            return hip_fail(ctx, e, "hipMalloc(arena plan, msm lane)");
        }
        a.cap = (size_t)caps10[i];
    }
    if (L.pinned_cap < pinned_cap) {
        if (L.pinned) hipHostFree(L.pinned);
        L.pinned = nullptr, L.pinned_cap = 0;
        const hipError_t e = hipHostMalloc(&L.pinned, (size_t)pinned_cap, hipHostMallocDefault);
        if (e != hipSuccess) {
            L.pinned = nullptr;
            return hip_fail(ctx, e, "hipHostMalloc(arena plan, msm lane)");
        }
        L.pinned_cap = (size_t)pinned_cap;
    }
    return ZK_OK;
}
void msm_lanes_destroy(zk_ctx* ctx) {
    for (int i = 0; i < zk_ctx::kLanes; i++) {
        zk_ctx::MsmLane& L = ctx->lanes[i];
        // a job nobody waited for may still be running on this lane: drain it before its arenas go
        if (L.main && i > 0) hipStreamSynchronize(L.main);
        for (auto& s2 : L.aux)
            if (s2) hipStreamSynchronize(s2);
        if (i > 0 && L.main) hipStreamDestroy(L.main);
        for (auto& s2 : L.aux)
            if (s2) hipStreamDestroy(s2);
        if (L.ev_fork) hipEventDestroy(L.ev_fork);
        for (auto& e : L.ev_join)
            if (e) hipEventDestroy(e);
        for (auto& e : L.ev_part)
            if (e) hipEventDestroy(e);
        for (auto& e : L.ev_cls)
            if (e) hipEventDestroy(e);
        for (auto& e : L.ev_sort)
            if (e) hipEventDestroy(e);
        for (auto& a : L.mem)
            if (a.p) hipFree(a.p);
        if (L.pinned) hipHostFree(L.pinned);
        L = zk_ctx::MsmLane();
    }
}

#endif
template <class Cv>
static int msm_enqueue(zk_ctx* ctx, MsmRun& run) {
    const MsmItem* items = run.items.data();
    const size_t count = run.count;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<MsmClass>& classes = run.classes;
    const Tuning& tn = tuning();
    if (tn.msm_share > 0 && tn.msm_share < 100) {  // experiment knob: a grid of persistent workgroups that leaves the other slots free
        // resident workgroups of k_accum_tiles<Cv> per CU (one static per instantiation; party threads may enqueue concurrently)
        static const int occ = [] {
            int o = 0;
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)k_accum_tiles<Cv>, kBlk, 0) == hipSuccess && o > 0 ? o : 2;
        }();
        run.accum_wg_cap = std::max<size_t>(1, (size_t)ctx->cu_count * (size_t)std::max(occ, 1) * (size_t)tn.msm_share / 100);
    }
    const u32 T_env = (u32)tn.msm_tile;
    const bool pair_env = tn.msm_pair != 0;
    const size_t fixq_max = (size_t)tn.msm_fixq;  // buckets per class
    const size_t quad_max = (size_t)tn.msm_quad;  // additions per pass (above it the plain pass is faster: measured)
    const int stage_env = (int)tn.msm_stage;
    const int split_env = (int)tn.msm_split;  // measured on MI355X: no gain (every phase is ALU-bound), off by default
    // ---- validate + classify by window width ----
    for (size_t k = 0; k < count; k++) {
        const MsmItem& it = items[k];
        if (!it.srs) return fail(ctx, ZK_ERR_INVALID, "null srs");
        if (it.srs->g2 != !Cv::kEndo) return fail(ctx, ZK_ERR_INVALID, "msm: the SRS belongs to the other group (G1 / G2)");
        if (it.offset + it.n > it.srs->n)
            return fail(ctx, ZK_ERR_LENGTH, "msm: %zu scalars but only %zu bases from offset %zu", it.n,
                        it.srs->n - std::min(it.offset, it.srs->n), it.offset);
        if (it.n >= ((size_t)1 << 30)) return fail(ctx, ZK_ERR_INVALID, "msm: n too large");  // 2n row entries, 31-bit indices
        if (it.n == 0) {
            run.empty_items.push_back(k);
            continue;
        }
        const bool shared = it.srs->d_table != nullptr && ctx->msm_window_override <= 0;
        int c = ctx->msm_window_override > 0 ? ctx->msm_window_override : msm_pick_window(it.n);
        if (count > 1 && ctx->msm_window_override <= 0) c = quantised_window(c);
        if (shared) c = it.srs->table_c;
        // window-table items of one table width can differ 8x in length (17-bit tables serve 2^15 .. 2^18 points): rows are as
        // long as the class's longest item, so such items get a class per power of two (padding < 2x; the digits / sort
        // passes walk the padded rows)
        int lgn = 0;
        while (((size_t)1 << lgn) < it.n) lgn++;
        const int size_key = (shared && tn.msm_size_classes) ? std::max(lgn, (int)tn.msm_size_class_min) : 0;  // (items of up to 2^msm_size_class_min points share a class)
        MsmClass* cl = nullptr;
        for (auto& x : classes)
            if (x.key_c == c && x.shared == shared && x.size_key == size_key) cl = &x;
        if (!cl) {
            classes.emplace_back();
            cl = &classes.back();
            cl->key_c = c;
            cl->size_key = size_key;
            cl->shared = shared;
            cl->L = msm_layout(c, (shared || !Cv::kEndo) ? kFullBits : kEndoBits);
            // the balanced layout may end up narrower than asked: buckets and planes follow the WIDEST window;
            // windows one bit narrower (index >= L.rem) use only the lower half of their bucket row
            cl->c = cl->L.base + (cl->L.rem ? 1 : 0);
            cl->narrow_from = (shared || cl->L.rem == 0) ? 0x7fffffff : cl->L.rem;
            cl->w0 = 0;
            cl->wc = cl->L.W;
            cl->nb = (size_t)1 << (cl->c - 1);
            cl->pair = pair_env;
            cl->npair = cl->pair ? std::max(1, cl->c / 2) : cl->c;
        }
        cl->idx.push_back(k);
        cl->ns = std::max(cl->ns, (it.n + 3) & ~(size_t)3);
    }
    if (classes.empty()) return ZK_OK;
    // ---- staggering: a class with enough work is cut by windows into parts (highest windows first).
    // The accumulation of part k+1 starts when that of part k ends, so the latency-bound tail of part k
    // (fix-up, bucket reduction, its piece of the host chain) and the sort of part k+1 hide behind the
    // ALU-bound accumulation; only the last part's tail stays exposed. ----
    {
        // (only the class with the most work is cut: one set of part events per ctx)
        size_t best = 0, best_work = 0;
        for (size_t i = 0; i < classes.size(); i++) {
            size_t work = 0;
            for (size_t k : classes[i].idx) work += items[k].n;
            if (work > best_work) best_work = work, best = i;
        }
        std::vector<MsmClass> cut;
        for (size_t i = 0; i < classes.size(); i++) {
            auto& cl = classes[i];
            const int want_parts = std::min(split_env, (int)zk_ctx::kParts);
            int parts = (i != best || cl.shared || want_parts < 2 || best_work < ((size_t)1 << 18) || cl.L.W < 2 * want_parts) ? 1 : want_parts;
            int hi = cl.L.W;
            for (int part = 0; part < parts; part++) {
                MsmClass x = cl;
                const int lo = cl.L.W * (parts - 1 - part) / parts;
                x.w0 = lo;
                x.wc = hi - lo;
                x.part = part;
                x.nparts = parts;
                hi = lo;
                cut.push_back(std::move(x));
            }
        }
        classes.swap(cut);
    }
    // ---- geometry per class, scratch high-water marks ----
    size_t need[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pinned_bytes = 0;
    const bool share_l1 = tn.msm_share_l1 != 0;
    for (auto& cl : classes) {
        const int W = cl.wc;
        const size_t nitems = cl.idx.size();
        cl.rpi = cl.shared ? 1 : W;
        cl.copies = cl.shared ? W : (Cv::kEndo ? 2 : 1);
        cl.row_len = (size_t)cl.copies * cl.ns;
        cl.rows = nitems * cl.rpi;
        cl.total = cl.rows * cl.nb;
        size_t nmax = 0;
        for (size_t k : cl.idx) nmax = std::max(nmax, items[k].n);
        // sorted entries per lane in k_accum_tiles: 32 when there is enough work to fill the chip;
        // for small MSMs balance the serial chain of a lane (T mixed adds, ~10 Fq-mul each) against
        // the fix-up chain of a bucket (entries_per_bucket / T full adds, ~14 Fq-mul each)
        cl.T = 32;
        // entries per bucket in a typical row: most rows are L.base bits wide, i.e. 2^(base-1) buckets in use
        const double per_bucket = (double)nmax * cl.copies / (double)(cl.shared ? cl.nb : std::min(cl.nb, (size_t)1 << (cl.L.base > 0 ? cl.L.base - 1 : 0)));
        if ((size_t)nitems * cl.rpi * cl.copies * nmax / 32 < (size_t)ctx->cu_count * 4 * 64 * 2) {
            cl.T = 4;
            while (cl.T < 32 && (double)cl.T * cl.T < 1.4 * per_bucket) cl.T <<= 1;
        } else {
            // long buckets: longer tiles keep the fix-up chain (tiles per bucket) at 2-3 while the lanes
            // still fill the chip (2 waves per SIMD)
            while (cl.T < 1024 && (double)cl.T * 2 <= per_bucket &&
                   (size_t)nitems * cl.rpi * cl.copies * nmax / (2 * cl.T) >= (size_t)ctx->cu_count * 4 * 64 * 2)
                cl.T <<= 1;
        }
        if (T_env) cl.T = T_env;
        cl.tiles_per_w = (cl.row_len + cl.T - 1) / cl.T;
        cl.total_tiles = cl.tiles_per_w * cl.rows;
        // sort geometry: 256..1024 partitions per row (top bits of the bucket), chunks of >= 16 Ki entries,
        // at most 512 chunks per row
        {
            // 256 partitions per row; more for very long rows so a partition stays near 16 Ki entries (its
            // level-2 workgroup and the region it scatters into stay small)
            size_t want = 256;
            const size_t np_env = (size_t)tn.msm_np;
            while (want < kMaxParts && cl.row_len / want > 16384) want <<= 1;
            if (np_env) want = np_env;
            while (want < kMaxParts && cl.nb / want > kMaxLow) want <<= 1;  // (tables wider than 20 bits: a partition holds at most kMaxLow buckets)
            cl.np = (u32)std::min<size_t>(cl.nb, want);
        }
        cl.low_bits = 0;
        while (((size_t)cl.np << cl.low_bits) < cl.nb) cl.low_bits++;
        {
            int ib = 1;
            while (((size_t)1 << ib) < cl.row_len) ib++;
            cl.idx_bits = (ib + cl.low_bits <= 31) ? ib : 0;
        }
        cl.staged_scatter = stage_env >= 0 ? stage_env != 0 : cl.row_len >= ((size_t)1 << 20);
        if (cl.rows * (size_t)cl.np * (cl.row_len / kStageChunk + 1) > ((size_t)64 << 20)) cl.staged_scatter = false;  // histogram arena <= 256 MiB
        cl.chunk_len = cl.staged_scatter ? kStageChunk : (std::max<size_t>(16384, (cl.row_len + 511) / 512) + 3) & ~(size_t)3;
        cl.nchunks = (u32)((cl.row_len + cl.chunk_len - 1) / cl.chunk_len);
        // round 6 (see k_tab_hist): long window-table rows are partitioned straight from the scalars, long partitions are sorted in tiles
        cl.fused_tab = cl.shared && cl.w0 == 0 && cl.wc == cl.L.W && cl.L.W <= kTabW && cl.np == kMaxParts && cl.row_len >= ((size_t)1 << (tn.msm_fused_min > 0 ? tn.msm_fused_min : 23)) &&
                       tn.msm_fused_min >= 0;
        if (cl.fused_tab) {
            cl.tab_spt = cl.L.W <= 16 ? 2 : 1;  // the stage holds tab_spt * 1024 * W words of LDS
            if (tn.msm_tab_spt == 1) cl.tab_spt = 1;  // (A/B: 66 KB of LDS, two workgroups per CU, runs half as long)
            cl.tab_c = tab_width_known(cl.key_c);
            if (cl.tab_c) {  // (the compile-time layout must be the class's)
                const WinLayout chk = msm_layout(cl.tab_c, kFullBits);
                if (chk.W != cl.L.W || chk.base != cl.L.base || chk.rem != cl.L.rem) cl.tab_c = 0;
            }
            const size_t chunk_sc = (size_t)cl.tab_spt * kSortThreads;
            const size_t nch = (cl.ns + chunk_sc - 1) / chunk_sc;
            if (cl.rows * (size_t)cl.np * nch > ((size_t)64 << 20)) cl.fused_tab = false;  // histogram arena <= 256 MiB
            else cl.nchunks = (u32)nch;
        }
        // (below ~2^23 entries per row the four extra launches cost what the run stores save: measured 2^16 .. 2^20 points, profiles/r06d_sort_ab.txt)
        cl.tiled_l2 = tn.msm_l2_tiled >= 0 && cl.row_len / cl.np >= (size_t)(tn.msm_l2_tiled > 0 ? tn.msm_l2_tiled : 8192) && (tn.msm_l2_tiled > 0 || cl.row_len >= ((size_t)1 << 23));
        cl.l2_ub = (u32)(cl.row_len / kL2Tile + cl.np);
        if (cl.tiled_l2 && cl.rows * (size_t)cl.l2_ub * ((size_t)1 << cl.low_bits) > ((size_t)64 << 20)) cl.tiled_l2 = false;  // tile histograms <= 256 MiB
        cl.cc_elems = cl.rows * (size_t)cl.np * cl.nchunks + 2 * cl.rows * (size_t)cl.np + cl.rows;  // hist, total, base, rowtot
        if (cl.tiled_l2) cl.cc_elems += cl.rows * (size_t)(cl.np + 1) + cl.rows * (size_t)cl.l2_ub * ((size_t)1 << cl.low_bits);  // tstart, h2
        // classes run concurrently on separate streams: each gets its own region of every arena
        const size_t total64 = (cl.total + 63) & ~(size_t)63, tiles64 = (cl.total_tiles + 63) & ~(size_t)63;  // XYZZ arrays: blocks of 64
        const size_t want_b[10] = {cl.rows * cl.row_len * 4, cl.rows * cl.row_len * 4, 2 * cl.total * 4 + cl.total_tiles * 4, total64 * Cv::kXyzzBytes, total64 * Cv::kXyzzBytes,
                                   2 * tiles64 * Cv::kXyzzBytes, (cl.total_tiles / kLongSpan + 64 + 1) * 4, nitems * sizeof(ItemDesc),
                                   cl.cc_elems * 4, cl.rows * cl.row_len * 2};
        for (int i = 0; i < 10; i++) {
            // the level-1 entries (part_idx, part_low: 6 of the ~10 bytes per entry) are dead once a class is sorted, long before its
            // accumulation ends: ONE region serves every class of the batch, and the sort phases run one after the other (event chain
            // below; they are bandwidth-bound and would share the memory system anyway) while the accumulations overlap as before.
            // n = 24 proof: 112 -> ~85 GB of pass arenas.
            const bool shared_scratch = share_l1 && (i == 1 || i == 9);
            cl.off[i] = shared_scratch ? 0 : need[i];
            if (shared_scratch) need[i] = std::max(need[i], (want_b[i] + 255) & ~(size_t)255);
            else need[i] += (want_b[i] + 255) & ~(size_t)255;
        }
        cl.pinned_off = pinned_bytes;
        pinned_bytes += ((cl.rows * (size_t)cl.npair * Cv::kJacBytes + 255) & ~(size_t)255) + ((nitems * sizeof(ItemDesc) + 255) & ~(size_t)255);
    }
    const bool dbg_classes = tn.msm_debug != 0;
    if (dbg_classes)
        for (auto& cl : classes) {
            size_t nmin = ~(size_t)0, nmax2 = 0, tot = 0;
            for (size_t k : cl.idx) nmin = std::min(nmin, items[k].n), nmax2 = std::max(nmax2, items[k].n), tot += items[k].n;
            fprintf(stderr, "zk-class key_c=%d c=%d W=%d items=%zu n=[%zu..%zu] sum=%zu ns=%zu rows=%zu nb=%zu buckets=%zu T=%u tiles=%zu np=%u\n", cl.key_c, cl.c, cl.wc,
                    cl.idx.size(), nmin, nmax2, tot, cl.ns, cl.rows, cl.nb, cl.total, cl.T, cl.total_tiles, cl.np);
        }
    // allocate every arena once, before anything is enqueued (no reallocation between classes)
    static const int slot[10] = {0, 1, 2, 3, 4, 5, 6, 9, 8, 11};
    void** buf = run.buf;
    run.pinned_bytes = pinned_bytes;
    if (run.own_mem) {
        // an asynchronous job takes a free lane: the one whose arenas already hold the job with the least slack, else the
        // roomiest one (its arenas grow once and stay: a prover issues the same batches proof after proof)
        int pick = -1;
        size_t pick_cap = 0;
        bool pick_fits = false;
        for (int l = 1; l < zk_ctx::kLanes; l++) {
            const zk_ctx::MsmLane& cand = ctx->lanes[l];
            if (cand.busy) continue;
            size_t cap = 0;
            bool fits = cand.pinned_cap >= pinned_bytes;
            for (int i = 0; i < 10; i++) cap += cand.mem[i].cap, fits = fits && cand.mem[i].cap >= need[i];
            const bool better = pick < 0 || (fits && !pick_fits) || (fits == pick_fits && (fits ? cap < pick_cap : cap > pick_cap));
            if (better) pick = l, pick_cap = cap, pick_fits = fits;
        }
        if (pick < 0) return fail(ctx, ZK_ERR_INVALID, "too many asynchronous MSM jobs in flight (at most %d): wait for one first", zk_ctx::kLanes - 1);
        run.lane = pick;
        {
            const int rc = msm_lane_prepare(ctx, run.lane);
            if (rc) return rc;
        }
        zk_ctx::MsmLane& LL = ctx->lanes[run.lane];
        for (int i = 0; i < 10; i++) {
            zk_ctx::Arena& a = LL.mem[i];
            if (a.cap < need[i] || !a.p) {
                if (a.p) hipFree(a.p);  // (the lane is idle: nothing can still use it)
                a.p = nullptr, a.cap = 0;
                const size_t want = std::max<size_t>(need[i], 256);
                const hipError_t e = device_alloc(ctx, &a.p, want);
                if (e != hipSuccess) {
                    a.p = nullptr;
                    return hip_fail(ctx, e, "hipMalloc(msm job arena)");
                }
                a.cap = want;
            }
            buf[i] = a.p;
        }
        if (LL.pinned_cap < pinned_bytes || !LL.pinned) {
            if (LL.pinned) hipHostFree(LL.pinned);
            LL.pinned = nullptr, LL.pinned_cap = 0;
            const size_t want = std::max<size_t>(pinned_bytes, 4096);
            const hipError_t e = hipHostMalloc(&LL.pinned, want, hipHostMallocDefault);
            if (e != hipSuccess) {
                LL.pinned = nullptr;
                return hip_fail(ctx, e, "hipHostMalloc(msm job)");
            }
            LL.pinned_cap = want;
        }
        run.hpin = (char*)LL.pinned;
        LL.busy = true;
    } else {
        run.lane = 0;
        {
            const int rc = msm_lane_prepare(ctx, 0);
            if (rc) return rc;
        }
        for (int i = 0; i < 10; i++) {
            buf[i] = scratch(ctx, slot[i], need[i]);
            if (!buf[i]) return ZK_ERR_OOM;
        }
        run.hpin = (char*)pinned(ctx, pinned_bytes);
        if (!run.hpin) return ZK_ERR_OOM;
    }
    zk_ctx::MsmLane& L = ctx->lanes[run.lane];
    const bool timers = run.lane == 0;
    char* hpin = run.hpin;
    if (run.lane != 0) {  // the scalars are produced by work on the ctx stream
        hipEventRecord(ctx->ev_async_in, ctx->stream);
        hipStreamWaitEvent(L.main, ctx->ev_async_in, 0);
    }
    // ---- enqueue every class without host synchronisation; independent classes go to separate
    // streams (the small ones are pure launch/latency chains and overlap with the big one).
    // Phase timers: sort of the first class, accumulation from the first part's launch to the last
    // part's end, then fix-up / reduction of the last part -- i.e. the exposed time of each phase. ----
    const bool serial_env = tn.msm_serial != 0;  // diagnostics: all classes on the ctx stream (per-kernel times = work)
    const bool multi = classes.size() > 1 && !serial_env;
    if (multi) {
        hipEventRecord(L.ev_fork, L.main);
        for (int k = 0; k < zk_ctx::kAux; k++) hipStreamWaitEvent(L.aux[k], L.ev_fork, 0);
    }
    // an error while classes are in flight: forked aux streams may still read the scratch arenas and the
    // caller's scalars -- drain the device before handing control (and those buffers) back
#define ZK_HIP_INFLIGHT(ctx, call)                                   \
    do {                                                             \
        hipError_t _e = (call);                                      \
        if (_e != hipSuccess) {                                      \
            hipDeviceSynchronize();                                  \
            msm_release(ctx, run);                                   \
            return zk::hip_fail(ctx, _e, #call);                     \
        }                                                            \
    } while (0)
    size_t cls_i = 0;
    for (auto& cl : classes) {
        const size_t nitems = cl.idx.size(), ns = cl.ns, nb = cl.nb, total = cl.total;
        const bool t_first = timers && (cls_i == 0), t_last = timers && (cls_i + 1 == (size_t)classes[0].nparts);  // the first class's parts carry the timers
        hipStream_t st = (!multi || cls_i == 0) ? L.main : L.aux[(cls_i - 1) % zk_ctx::kAux];
        u32* digits = (u32*)((char*)buf[0] + cl.off[0]);
        u32* sorted = digits;  // the digits are dead once partitioned: the sorted entries take their place
        u32* part_idx = (u32*)((char*)buf[1] + cl.off[1]);
        unsigned short* part_low = (unsigned short*)((char*)buf[9] + cl.off[9]);
        uint2* oc = (uint2*)((char*)buf[2] + cl.off[2]);  // (offset, count) per bucket
        u32* tile_b = (u32*)(oc + total);                // first bucket of every tile
        void* bufA = (char*)buf[3] + cl.off[3];
        void* bufB = (char*)buf[4] + cl.off[4];
        void* heads = (char*)buf[5] + cl.off[5];
        void* tails = (char*)heads + ((cl.total_tiles + 63) & ~(size_t)63) * Cv::kXyzzBytes;
        u32* longs = (u32*)((char*)buf[6] + cl.off[6]);
        ItemDesc* d_items = (ItemDesc*)((char*)buf[7] + cl.off[7]);
        u32* cc = (u32*)((char*)buf[8] + cl.off[8]);
        uint64_t* h_pts = (uint64_t*)(hpin + cl.pinned_off);
        ItemDesc* h_items = (ItemDesc*)(hpin + cl.pinned_off + ((cl.rows * (size_t)cl.npair * Cv::kJacBytes + 255) & ~(size_t)255));
        for (size_t j = 0; j < nitems; j++) {
            const MsmItem& it = items[cl.idx[j]];
            h_items[j].scalars = it.d_scalars;
            h_items[j].rec = cl.shared ? (u32)it.srs->table_rec : (u32)Cv::kAffBytes;
            h_items[j].bases = (const char*)(cl.shared ? it.srs->d_table : it.srs->d_bases) + it.offset * (size_t)h_items[j].rec;
            h_items[j].n = (u32)it.n;
            h_items[j].pstride = cl.shared ? (u32)it.srs->table_stride : (Cv::kEndo ? (u32)it.srs->n : 0u);  // phi(P_i) sits n points after P_i
        }
        if (t_first) hipEventRecord(ctx->ev[0], st);
        ZK_HIP_INFLIGHT(ctx, hipMemcpyAsync(d_items, h_items, nitems * sizeof(ItemDesc), hipMemcpyHostToDevice, st));
        ZK_HIP_INFLIGHT(ctx, hipMemsetAsync(longs, 0, 4, st));
        if (share_l1) {
            while (L.ev_sort.size() <= cls_i) {
                hipEvent_t e = nullptr;
                ZK_HIP_INFLIGHT(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                L.ev_sort.push_back(e);
            }
            if (cls_i > 0) hipStreamWaitEvent(st, L.ev_sort[cls_i - 1], 0);  // the shared level-1 scratch is free again
        }
        if (!cl.fused_tab)
            hipLaunchKernelGGL(k_digits, dim3((unsigned)((ns + kBlk - 1) / kBlk), (unsigned)nitems), dim3(kBlk), 0, st,
                               (const ItemDesc*)d_items, ns, cl.L, cl.w0, cl.wc, (cl.shared || !Cv::kEndo) ? 0 : 1, digits);
        {
            u32* hist = cc;
            u32* ptotal = hist + cl.rows * (size_t)cl.np * cl.nchunks;
            u32* pbase = ptotal + cl.rows * (size_t)cl.np;
            u32* rowtot = pbase + cl.rows * (size_t)cl.np;
            const dim3 g_chunks((unsigned)(cl.nchunks * cl.rows)), g_parts((unsigned)(cl.np * cl.rows));
            const RowReal rr{(const ItemDesc*)d_items, (u32)cl.rpi, (u32)ns};
            if (cl.fused_tab)
                tab_launch_hist(cl.tab_c, g_chunks, st, (const ItemDesc*)d_items, cl.L, cl.tab_spt * (u32)kSortThreads, cl.nchunks, cl.np, cl.low_bits, hist);
            else
                hipLaunchKernelGGL(k_part_hist, g_chunks, dim3(kSortThreads), 0, st, (const u32*)digits, cl.row_len, cl.chunk_len, cl.nchunks, cl.np,
                                   cl.low_bits, hist, rr);
            hipLaunchKernelGGL(k_part_scan, dim3((unsigned)(cl.rows * cl.np)), dim3(cl.nchunks >= 2048 ? 1024 : 128), 0, st, hist, cl.nchunks, ptotal);
            hipLaunchKernelGGL(k_part_bases, dim3((unsigned)cl.rows), dim3(cl.np <= 256 ? 256 : kScanThreads), 0, st, (const u32*)ptotal, cl.np, pbase, rowtot);
            if (cl.fused_tab) {
                const size_t lds = (3 * (size_t)kMaxParts + 1 + 160 + (size_t)cl.tab_spt * kSortThreads * cl.L.W) * 4;
                tab_launch_scatter(cl.tab_c, (int)cl.tab_spt, dim3((unsigned)(8 * ((cl.nchunks + 7) / 8) * cl.rows)), lds, st, (const ItemDesc*)d_items, (u32)ns, cl.L, cl.nchunks, cl.np, cl.low_bits, (const u32*)hist,
                                   (const u32*)ptotal, (const u32*)pbase, cl.idx_bits, cl.row_len, part_idx, part_low);
            } else if (cl.staged_scatter) {
                const size_t lds = (3 * (size_t)kMaxParts + kSortThreads + 2 * (size_t)kStageChunk) * 4;
                hipFuncSetAttribute((const void*)k_part_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL(k_part_scatter_staged, g_chunks, dim3(kSortThreads), lds, st, (const u32*)digits, cl.row_len, cl.chunk_len,
                                   cl.nchunks, cl.np, cl.low_bits, (const u32*)hist, (const u32*)pbase, cl.idx_bits, part_idx, part_low, rr);
            } else {
                hipLaunchKernelGGL(k_part_scatter, g_chunks, dim3(kSortThreads), 0, st, (const u32*)digits, cl.row_len, cl.chunk_len, cl.nchunks,
                                   cl.np, cl.low_bits, (const u32*)hist, (const u32*)pbase, cl.idx_bits, part_idx, part_low, rr);
            }
            if (cl.tiled_l2) {
                u32* tstart = rowtot + cl.rows;
                u32* h2 = tstart + cl.rows * (size_t)(cl.np + 1);
                const u32 ub = cl.l2_ub;
                const dim3 g_tiles((unsigned)(ub * cl.rows));
                const size_t lds = (2 * ((size_t)1 << cl.low_bits) + 1 + 160 + kL2Tile) * 4;
                hipLaunchKernelGGL(k_l2_tiles, dim3((unsigned)cl.rows), dim3(kScanThreads), 0, st, (const u32*)pbase, (const u32*)rowtot, cl.np, tstart);
                hipLaunchKernelGGL(k_l2_hist, g_tiles, dim3(kSortThreads), 0, st, (const u32*)part_idx, (const unsigned short*)part_low, cl.row_len, cl.np, cl.low_bits, cl.idx_bits,
                                   (const u32*)pbase, (const u32*)rowtot, (const u32*)tstart, ub, h2);
                hipLaunchKernelGGL(k_l2_scan, g_parts, dim3(kSortThreads), 0, st, cl.np, cl.low_bits, nb, (const u32*)pbase, (const u32*)tstart, ub, h2, oc, cl.T, cl.tiles_per_w, tile_b);
                hipFuncSetAttribute((const void*)k_l2_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL(k_l2_scatter, dim3((unsigned)(8 * ((ub + 7) / 8) * cl.rows)), dim3(kSortThreads), lds, st, (const u32*)part_idx, (const unsigned short*)part_low, cl.row_len, cl.np, cl.low_bits,
                                   cl.idx_bits, nb, (const u32*)pbase, (const u32*)rowtot, (const u32*)tstart, ub, (const u32*)h2, (const uint2*)oc, sorted);
            } else
            hipLaunchKernelGGL(k_part_sort, g_parts, dim3(cl.row_len / cl.np >= 4096 ? kSortThreads : 256), 0, st, (const u32*)part_idx, (const unsigned short*)part_low, cl.row_len,
                               cl.np, cl.low_bits, cl.idx_bits, nb, (const u32*)pbase, (const u32*)rowtot, oc, cl.T, cl.tiles_per_w, tile_b, sorted);
        }
        if (share_l1) hipEventRecord(L.ev_sort[cls_i], st);
        if (t_first) hipEventRecord(ctx->ev[1], st);
        if (cl.part > 0) hipStreamWaitEvent(st, L.ev_part[(cl.part - 1) % zk_ctx::kParts], 0);  // after the previous part's accumulation
        // (a pass with a share below 100 % leaves workgroup slots free for the kernels of other streams: knob msm_share)
        const size_t accum_wgs_full = (cl.total_tiles + kBlk - 1) / kBlk;
        const size_t accum_wgs = run.accum_wg_cap ? std::min<size_t>(accum_wgs_full, run.accum_wg_cap) : accum_wgs_full;
        hipLaunchKernelGGL((k_accum_tiles<Cv>), dim3((unsigned)accum_wgs), dim3(kBlk), 0, st,
                           (const ItemDesc*)d_items, cl.rpi, (const u32*)sorted, (const uint2*)oc, (const u32*)tile_b, cl.row_len,
                           (u32)ns, ((ns & (ns - 1)) == 0) ? (int)__builtin_ctzll(ns) : -1, nb, cl.T, cl.tiles_per_w, cl.total_tiles, bufA, heads, tails, (int)tn.msm_idx_ahead);
        if (cl.nparts > 1 && cl.part + 1 < cl.nparts) hipEventRecord(L.ev_part[cl.part % zk_ctx::kParts], st);
        if (t_last) hipEventRecord(ctx->ev[4], st);
        const NarrowRows nrw{cl.rpi, cl.w0, cl.narrow_from};
        if (Cv::kQuad && total <= fixq_max)
            hipLaunchKernelGGL((k_fixup_quad<Cv>), dim3((unsigned)((4 * total + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, (const uint2*)oc,
                               nb, nrw, cl.T, cl.tiles_per_w, total, bufA, (const void*)heads, (const void*)tails, longs,
                               longs + 1);
        else
            hipLaunchKernelGGL((k_fixup<Cv>), dim3((unsigned)((total + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, (const uint2*)oc,
                               nb, nrw, cl.T, cl.tiles_per_w, total, bufA, (const void*)heads, (const void*)tails, longs,
                               longs + 1);
        hipLaunchKernelGGL((k_fixup_long<Cv>), dim3(512), dim3(Cv::kXyzzBytes == 192 ? kBlk : kBlk / 2), (size_t)48 * 1024, st, (const uint2*)oc, nb, cl.T,
                           cl.tiles_per_w, bufA, (const void*)heads, (const void*)tails, (const u32*)longs, (const u32*)(longs + 1));
        if (t_last) hipEventRecord(ctx->ev[2], st);
        void* in = bufA;
        void* out = bufB;
        int rows = 1;
        size_t len = nb;
        while (len > 1) {
            size_t threads = cl.rows * (size_t)(rows + 1) * (len >> 1);
            if (Cv::kQuad && threads <= quad_max)  // too few additions to fill the chip: spend four lanes on each
                hipLaunchKernelGGL((k_halve_quad<Cv>), dim3((unsigned)((4 * threads + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, (const void*)in, out,
                                   (int)cl.rows, rows, len, nrw);
            else
                hipLaunchKernelGGL((k_halve<Cv>), dim3((unsigned)((threads + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, (const void*)in, out,
                                   (int)cl.rows, rows, len, nrw);
            std::swap(in, out);
            rows++;
            len >>= 1;
        }
        // rows == c reduced points per window row -> Jacobian points in the reference form
        {
            const size_t threads = cl.rows * (size_t)cl.npair;
            hipLaunchKernelGGL((k_finish<Cv>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, (const void*)in, out, cl.rows, cl.c, cl.npair, cl.pair ? 1 : 0);
        }
        ZK_HIP_INFLIGHT(ctx, hipGetLastError());
        ZK_HIP_INFLIGHT(ctx, hipMemcpyAsync(h_pts, out, cl.rows * (size_t)cl.npair * Cv::kJacBytes, hipMemcpyDeviceToHost, st));
        if (t_last) hipEventRecord(ctx->ev[3], st);
        while (L.ev_cls.size() <= cls_i) {  // one completion event per class: the host starts on a class as soon as it lands
            hipEvent_t e;
            ZK_HIP_INFLIGHT(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            L.ev_cls.push_back(e);
        }
        hipEventRecord(L.ev_cls[cls_i], st);
        cls_i++;
    }
#undef ZK_HIP_INFLIGHT
    if (multi) {  // join: later work on the lane's main stream is ordered after every class
        for (int k = 0; k < zk_ctx::kAux; k++) {
            hipEventRecord(L.ev_join[k], L.aux[k]);
            hipStreamWaitEvent(L.main, L.ev_join[k], 0);
        }
    }
    return ZK_OK;
}

template <class Cv>
static int msm_finish(zk_ctx* ctx, MsmRun& run, uint64_t* h_out) {
    typedef typename Cv::HostF HF;
    constexpr size_t kOutWords = 3 * zkhost::fe_words<HF>();  // 18 (G1) / 36 (G2) u64 per result
    const size_t count = run.count;
    std::vector<MsmClass>& classes = run.classes;
    zk_ctx::MsmLane& L = ctx->lanes[run.lane];
    char* hpin = run.hpin;
    for (size_t k : run.empty_items) zkhost::write_normalised(zkhost::jac_inf_t<HF>(), h_out + kOutWords * k);
    if (classes.empty()) return ZK_OK;
    // ---- host combine: one doubling chain of ~129 steps per item, run by the ctx's worker pool.  Classes
    // are taken in the order they finish on the device (least work first; the parts of a staggered class
    // in window order): while the big class is still running, the items of the small ones are already
    // being combined.  Only the last class's chains are exposed. ----
    std::vector<zkhost::JacT<HF>> chain(count, zkhost::jac_inf_t<HF>());
    auto run_item = [&chain, hpin, h_out](const MsmClass& cl, size_t j) {
        WinLayout one{1, 0, 0};  // shared buckets: a single row, the window factors live in the table
        const uint64_t* h = (const uint64_t*)(hpin + cl.pinned_off) + j * (size_t)cl.rpi * cl.npair * kOutWords;
        if (cl.shared) combine_windows<HF>(chain[cl.idx[j]], h, one, 0, 1, cl.c, cl.npair, cl.pair);
        else combine_windows<HF>(chain[cl.idx[j]], h, cl.L, cl.w0, cl.wc, cl.c, cl.npair, cl.pair);
        if (cl.part + 1 == cl.nparts) zkhost::write_normalised(chain[cl.idx[j]], h_out + kOutWords * cl.idx[j]);
    };
    std::vector<size_t> order(classes.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        const MsmClass &x = classes[a], &y = classes[b];
        if (x.nparts > 1 || y.nparts > 1) return a < b;  // staggered parts stay in enqueue (= window) order, last
        return x.rows * x.row_len < y.rows * y.row_len;
    });
    HostPool* pool = classes.size() > 1 || count > 1 ? host_pool(ctx) : nullptr;
    float host_ms = 0;
    for (size_t oi = 0; oi < order.size(); oi++) {
        const MsmClass& cl = classes[order[oi]];
        {
            const hipError_t e = hipEventSynchronize(L.ev_cls[order[oi]]);
            if (e != hipSuccess) {
                if (pool) pool->wait();  // jobs in flight reference this frame
                msm_release(ctx, run);
                return hip_fail(ctx, e, "hipEventSynchronize(class done)");
            }
        }
        auto t0 = std::chrono::steady_clock::now();
        if (cl.nparts > 1 && cl.part > 0 && pool) pool->wait();  // the previous part's chains must have advanced first
        for (size_t j = 0; j < cl.idx.size(); j++) {
            if (pool && (cl.idx.size() > 1 || oi + 1 < order.size())) pool->submit([&run_item, &cl, j] { run_item(cl, j); });
            else run_item(cl, j);
        }
        if (oi + 1 == order.size()) {
            if (pool) pool->wait();
            host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();  // exposed part
        }
    }
    {
        const hipError_t e = hipStreamSynchronize(L.main);
        msm_release(ctx, run);  // (also on failure: an async lane must not stay busy for the rest of the ctx's life)
        if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamSynchronize(msm lane)");
    }
    if (run.lane != 0) return ZK_OK;  // (the phase timers belong to the blocking calls)
    float ms;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
    ctx->msm_ms[0] = ms;  // digits + sort (first part)
    hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[4]);
    ctx->msm_ms[1] = ms;  // k_accum_tiles: first part's launch to last part's end (the dominant kernel)
    hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[2]);
    ctx->msm_ms[2] = ms;  // fix-up (last part)
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]);
    ctx->msm_ms[3] = ms;  // bucket reduction + D2H (last part)
    ctx->msm_ms[4] = host_ms;  // host combine after the device has finished
    ctx->msm_ms[5] = ctx->msm_ms[0] + ctx->msm_ms[1] + ctx->msm_ms[2] + ctx->msm_ms[3] + ctx->msm_ms[4];
    return ZK_OK;
}

template <class Cv>
static int msm_batch(zk_ctx* ctx, const MsmItem* items, size_t count, uint64_t* h_out) {
    if (!h_out && count) return fail(ctx, ZK_ERR_INVALID, "null argument");
    MsmRun run;
    run.count = count;
    run.items.assign(items, items + count);
    int rc = msm_enqueue<Cv>(ctx, run);
    if (rc) return rc;
    return msm_finish<Cv>(ctx, run, h_out);
}

#ifndef ZK_MSM_TU_G2
int msm_g1_batch(zk_ctx* ctx, const MsmItem* items, size_t count, uint64_t* h_out) { return msm_batch<CvG1>(ctx, items, count, h_out); }
#else
int msm_g2_batch(zk_ctx* ctx, const MsmItem* items, size_t count, uint64_t* h_out) { return msm_batch<CvG2>(ctx, items, count, h_out); }
// the G2 half of srs_precompute: table[w][i] = 2^{bit_offset(w)} P_i for a G2 level
int srs_precompute_table_g2(zk_ctx* ctx, const zk_srs* srs, int c, size_t nsr, void* d_table) {
    const WinLayout L = msm_layout(c, kFullBits);
    hipLaunchKernelGGL((k_precompute<CvG2>), dim3((unsigned)((srs->n + 63) / 64)), dim3(64), 0, ctx->stream, (const void*)srs->d_bases, srs->n, nsr, L, d_table, (u32)CvG2::kAffBytes);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}
#endif

}  // namespace zk
#ifndef ZK_MSM_TU_G2  // (G1 translation unit only: asynchronous jobs, SRS handling, host combinations, G1 test hooks)
struct zk_msm_job {
    zk::MsmRun run;
};
namespace zk {
int msm_g1_batch_async(zk_ctx* ctx, const MsmItem* items, size_t count, zk_msm_job** job) {
    if (!job) return fail(ctx, ZK_ERR_INVALID, "null argument");
    *job = nullptr;
    zk_msm_job* j = new zk_msm_job();
    j->run.count = count;
    j->run.items.assign(items, items + count);
    j->run.own_mem = true;  // (the lane is picked inside msm_enqueue, once the job's memory needs are known)
    const int rc = msm_enqueue<CvG1>(ctx, j->run);
    if (rc) {
        delete j;
        return rc;
    }
    *job = j;
    return ZK_OK;
}
int msm_job_wait(zk_ctx* ctx, zk_msm_job* job, uint64_t* h_out) {
    if (!job) return fail(ctx, ZK_ERR_INVALID, "null job");
    int rc;
    if (!h_out && job->run.count) {
        if (job->run.lane > 0) hipStreamSynchronize(ctx->lanes[job->run.lane].main);
        msm_release(ctx, job->run);
        rc = fail(ctx, ZK_ERR_INVALID, "null argument");
    } else {
        rc = msm_finish<CvG1>(ctx, job->run, h_out);
    }
    delete job;
    return rc;
}

int msm_g1(zk_ctx* ctx, const zk_srs* srs, size_t offset, const void* d_scalars, size_t n, uint64_t* h_out) {
    if (!srs || !h_out) return fail(ctx, ZK_ERR_INVALID, "null argument");
    MsmItem it{srs, offset, d_scalars, n};
    return msm_g1_batch(ctx, &it, 1, h_out);
}

// ---------------------------------------------------------------------------------------
static void srs_convert(zk_ctx* ctx, const void* in, void* out, size_t npoints, bool to_internal) {
    const size_t ncoord = 2 * npoints;
    size_t blocks = std::min<size_t>((ncoord + kBlk - 1) / kBlk, (size_t)ctx->cu_count * 8);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_srs_convert, dim3((unsigned)blocks), dim3(kBlk), 0, ctx->stream, in, out, ncoord, to_internal ? 1 : 0);
}

static void srs_endo(zk_ctx* ctx, zk_srs* s) {
    size_t blocks = std::min<size_t>((s->n + kBlk - 1) / kBlk, (size_t)ctx->cu_count * 8);
    if (blocks) hipLaunchKernelGGL(k_srs_endo, dim3((unsigned)blocks), dim3(kBlk), 0, ctx->stream, s->d_bases, s->n);
}

// frees a half-built zk_srs (and its device allocation) when a constructor leaves on an error path
struct SrsGuard {
    zk_srs* s = nullptr;
    ~SrsGuard() {
        if (!s) return;
        if (s->d_bases) hipFree(s->d_bases);
        delete s;
    }
    zk_srs* release() {
        zk_srs* r = s;
        s = nullptr;
        return r;
    }
};

// The device copy of an SRS is kept in the kernels' INTERNAL form (Montgomery radix 2^390, see
// fq30.cuh): one conversion pass at registration, like the reference's own `mature()` step.
int srs_pack(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out) {
    if (!out || (n && !h_bases)) return fail(ctx, ZK_ERR_INVALID, "null argument");
    if (stride != 96 && stride < 97) return fail(ctx, ZK_ERR_INVALID, "stride must be 96 or >= 97 (x, y, infinity flag)");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    SrsGuard guard;
    zk_srs* s = guard.s = new zk_srs();
    s->n = n;
    s->owned = true;
    if (n) {
        ZK_HIP(ctx, device_alloc(ctx, &s->d_bases, 2 * n * 96));  // P_i, then phi(P_i)
        if (stride == 96) {
            ZK_HIP(ctx, hipMemcpyAsync(s->d_bases, h_bases, n * 96, hipMemcpyHostToDevice, ctx->stream));
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            std::vector<char> packed(n * 96);
            const char* src = (const char*)h_bases;
            for (size_t i = 0; i < n; i++) {
                if (src[i * stride + 96]) std::memset(&packed[i * 96], 0, 96);  // infinity flag (ark Affine.infinity)
                else std::memcpy(&packed[i * 96], src + i * stride, 96);
            }
            ZK_HIP(ctx, hipMemcpy(s->d_bases, packed.data(), n * 96, hipMemcpyHostToDevice));
        }
        srs_convert(ctx, s->d_bases, s->d_bases, n, true);
        srs_endo(ctx, s);
        ZK_HIP(ctx, hipGetLastError());
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = guard.release();
    return ZK_OK;
}

// G2 base vector (powers_of_g2 and friends, dpoly_comm.rs:27): ark G2Affine = { x: Fq2, y: Fq2, infinity } -> 192-byte
// records x.c0 | x.c1 | y.c0 | y.c1 (stride 192, or the Rust struct's stride with the flag byte at offset 192)
int srs_pack_g2(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out) {
    if (!out || (n && !h_bases)) return fail(ctx, ZK_ERR_INVALID, "null argument");
    if (stride != 192 && stride < 193) return fail(ctx, ZK_ERR_INVALID, "stride must be 192 or >= 193 (x, y, infinity flag)");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    SrsGuard guard;
    zk_srs* s = guard.s = new zk_srs();
    s->n = n;
    s->owned = true;
    s->g2 = true;
    if (n) {
        ZK_HIP(ctx, device_alloc(ctx, &s->d_bases, n * 192));
        std::vector<char> packed;
        const void* src = h_bases;
        if (stride != 192) {
            packed.resize(n * 192);
            const char* p = (const char*)h_bases;
            for (size_t i = 0; i < n; i++) {
                if (p[i * stride + 192]) std::memset(&packed[i * 192], 0, 192);
                else std::memcpy(&packed[i * 192], p + i * stride, 192);
            }
            src = packed.data();
        }
        ZK_HIP(ctx, hipMemcpy(s->d_bases, src, n * 192, hipMemcpyHostToDevice));
        srs_convert(ctx, s->d_bases, s->d_bases, 2 * n, true);  // four coordinates per point
        ZK_HIP(ctx, hipGetLastError());
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = guard.release();
    return ZK_OK;
}

int srs_from_device(zk_ctx* ctx, const void* d_bases96, size_t n, zk_srs** out) {
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    SrsGuard guard;
    zk_srs* s = guard.s = new zk_srs();
    s->n = n;
    s->owned = true;
    if (n) {
        ZK_HIP(ctx, device_alloc(ctx, &s->d_bases, 2 * n * 96));  // P_i, then phi(P_i)
        srs_convert(ctx, d_bases96, s->d_bases, n, true);
        srs_endo(ctx, s);
        ZK_HIP(ctx, hipGetLastError());
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = guard.release();
    return ZK_OK;
}

int srs_download(zk_ctx* ctx, const zk_srs* srs, void* h_out96) {
    if (!srs || (srs->n && !h_out96)) return fail(ctx, ZK_ERR_INVALID, "null argument");
    if (!srs->n) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rec = srs->g2 ? 192 : 96;  // G2: x.c0 | x.c1 | y.c0 | y.c1
    void* tmp = scratch(ctx, 0, srs->n * rec);
    if (!tmp) return ZK_ERR_OOM;
    srs_convert(ctx, srs->d_bases, tmp, srs->n * (rec / 96), false);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(h_out96, tmp, srs->n * rec, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

int dbg_fq(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n) {
    if (n == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    size_t blocks = std::min<size_t>((n + kBlk - 1) / kBlk, (size_t)ctx->cu_count * 8);
    hipLaunchKernelGGL(k_dbg_fq, dim3((unsigned)blocks), dim3(kBlk), 0, ctx->stream, a, b, out, n, op);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

// window bits of a G2 level's table: an Fq2 addition costs ~3x a G1 one in the reduction passes as well (quad-lane form in
// both, CvG2::kQuad), so the optimum sits lower than G1's (tools/g2_time.py sweep on MI355X)
static int msm_pick_window_full_g2(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = lg <= 11 ? 13 : (lg <= 13 ? 15 : 16);  // 16 bits = exactly 16 windows: the optimum from 2^14 to 2^18 points
    c += (int)tuning().msm_table_dc;
    return std::max(4, std::min(20, c));
}

int srs_precompute_table_g2(zk_ctx* ctx, const zk_srs* srs, int c, size_t nsr, void* d_table);  // zk_msm.hip built with -DZK_MSM_TU_G2
int srs_precompute(zk_ctx* ctx, zk_srs* srs, int c, int record_bytes) {
    if (!srs) return fail(ctx, ZK_ERR_INVALID, "null srs");
    if (c == 0) c = srs->g2 ? msm_pick_window_full_g2(srs->n ? srs->n : 1) : msm_pick_window_full(srs->n ? srs->n : 1);
    if (c < 2 || c > kMaxTableBits) return fail(ctx, ZK_ERR_INVALID, "window bits out of range");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    if (srs->d_table) {
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(srs->d_table);
        srs->d_table = nullptr;
    }
    if (srs->n == 0) return ZK_OK;
    const WinLayout L = msm_layout(c, kFullBits);
    const size_t nsr = (srs->n + 3) & ~(size_t)3;
    // A G1 point is 96 B.  Packed, two records of three straddle a 128-B line and a gather of the accumulation moves 1.67 lines on average; one
    // record per 128-B line costs +33 % table memory.  Measured (profiles/r05zb_table_rec_ab.txt): k_accum_tiles at 2^20 2.04-2.14 -> 1.90-1.91 ms, the
    // 2^20 MSM 3.57-3.70e8 -> 3.82-3.84e8 scalar-muls/s, 2^24 4.69 -> 4.80e8, n = 24 proof 0.815 -> 0.793 s.
    // Round 6: the DEFAULT (record_bytes = 0 and the knob srs_table_rec = 0) decides per table by what the device can spare right now -- 128-B
    // records when the table then leaves >= 60 % of the device free, packed 96-B records otherwise: a prover's ~40 levels, or several parties
    // sharing one GPU, fall back to the packed form by themselves as the memory fills (the hosts' parameter sets add their own floor below which
    // a level gets no table at all: zkhost/hyperplonk.hpp finish_setup).  zk_srs_precompute_layout(.., 96 | 128) forces one form.
    long want_rec = record_bytes ? (long)record_bytes : tuning().srs_table_rec;
    if (want_rec == 0 && !srs->g2) {
        size_t fr = 0, tot = 0;
        want_rec = (hipMemGetInfo(&fr, &tot) == hipSuccess && (double)fr - (double)L.W * (double)nsr * 128.0 >= 0.6 * (double)tot) ? 128 : 96;
    }
    const size_t rec = srs->g2 ? CvG2::kAffBytes : (want_rec == 128 ? (size_t)128 : CvG1::kAffBytes);
    ZK_HIP(ctx, device_alloc(ctx, &srs->d_table, (size_t)L.W * nsr * rec));
    ZK_HIP(ctx, hipMemsetAsync(srs->d_table, 0, (size_t)L.W * nsr * rec, ctx->stream));
    if (srs->g2) {
        const int rc = srs_precompute_table_g2(ctx, srs, c, nsr, srs->d_table);  // (the other translation unit)
        if (rc) return rc;
    } else {
        // (tables of >= 2^10 points: the batched normalisation; below, and when its transient arrays do not fit, one inversion per record)
        int rc = (srs->n >= 1024 && tuning().srs_table_batched != 0) ? precompute_table_g1_batched(ctx, srs, L, nsr, srs->d_table, (u32)rec) : ZK_ERR_OOM;
        if (rc != ZK_OK && rc != ZK_ERR_OOM) return rc;
        if (rc == ZK_ERR_OOM)
            hipLaunchKernelGGL((k_precompute<CvG1>), dim3((unsigned)((srs->n + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)srs->d_bases, srs->n, nsr,
                               L, srs->d_table, (u32)rec);
    }
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    srs->table_c = c;
    srs->table_stride = nsr;
    srs->table_rec = rec;
    return ZK_OK;
}

void srs_finish_endo(zk_ctx* ctx, zk_srs* s) { srs_endo(ctx, s); }
void srs_convert_form(zk_ctx* ctx, const void* in, void* out, size_t npoints, bool to_internal) { srs_convert(ctx, in, out, npoints, to_internal); }

// zk_srs.hip: d_out96[i] = start + i * step in the internal affine form
int fill_sequence_affine(zk_ctx* ctx, const zkhost::Aff& start, const zkhost::Aff& step, size_t n, void* d_out96);

int srs_generate(zk_ctx* ctx, const uint64_t* k0, const uint64_t* k1, size_t n, zk_srs** out) {
    namespace H = zkhost;
    if (!out) return fail(ctx, ZK_ERR_INVALID, "null argument");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    H::Aff G{H::to_mont(H::GX_CANON), H::to_mont(H::GY_CANON)};
    const H::Aff start = H::jac_to_aff(H::scalar_mul(G, k0));
    const H::Aff step = H::jac_to_aff(H::scalar_mul(G, k1));
    SrsGuard guard;
    zk_srs* s = guard.s = new zk_srs();
    s->n = n;
    s->owned = true;
    if (n) {
        ZK_HIP(ctx, device_alloc(ctx, &s->d_bases, 2 * n * 96));  // P_i, then phi(P_i)
        int rc = fill_sequence_affine(ctx, start, step, n, s->d_bases);  // the walk and the normalisation run on the device
        if (rc) return rc;
        srs_endo(ctx, s);
        ZK_HIP(ctx, hipGetLastError());
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = guard.release();
    return ZK_OK;
}

// K9: tiny public linear maps on points (PSS unpack2 / pack of d_msm's leader closure,
// dmsm.rs:30-39; sums of N_p commitments dpoly_comm.rs:289-292): sum_i k_i * P_i for a handful
// of points.  A joint double-and-add on the host: it is a ~255-step dependency chain, the same
// shape as the MSM's final combine.
// count independent combinations sharing the scalar vector: out[r] = sum_i k_i * P[r][i].
// One batch inversion normalises all inputs that need it, one more all outputs.
int g1_lincomb_batch_host(zk_ctx* ctx, const uint64_t* h_points_jac, const uint64_t* h_scalars_canon, size_t n, size_t count,
                          uint64_t* h_out) {
    namespace H = zkhost;
    const size_t total = n * count;
    std::vector<H::Aff> pts(total);
    std::vector<H::Jac> tmp(total);
    bool all_affine = true;
    for (size_t i = 0; i < total; i++) {
        std::memcpy(tmp[i].x.data(), h_points_jac + 18 * i, 48);
        std::memcpy(tmp[i].y.data(), h_points_jac + 18 * i + 6, 48);
        std::memcpy(tmp[i].z.data(), h_points_jac + 18 * i + 12, 48);
        if (!(tmp[i].z == H::ONE) && !H::is_zero(tmp[i].z)) all_affine = false;
    }
    if (all_affine) {  // library outputs are already normalised: no inversion needed
        for (size_t i = 0; i < total; i++) pts[i] = H::is_zero(tmp[i].z) ? H::Aff{H::ZERO, H::ZERO} : H::Aff{tmp[i].x, tmp[i].y};
    } else if (total) {
        H::batch_to_affine(tmp, pts.data());
    }
    int top = -1;  // highest set bit over all scalars (coefficients 1 of d_commit / d_open sums: top = 0)
    for (size_t i = 0; i < n; i++)
        for (int b = 255; b > top; b--)
            if ((h_scalars_canon[4 * i + b / 64] >> (b % 64)) & 1) {
                top = b;
                break;
            }
    std::vector<H::Jac> acc(count, H::jac_inf());
    // rows are independent ~255-step chains: host threads over rows.  A row whose points are all the same
    // (the no-comm echo net hands the leader N copies of its own message) collapses to ONE scalar
    // multiplication by the sum of the coefficients.
    uint64_t ksum[4] = {0, 0, 0, 0};
    {
        H::u128 cy = 0;
        uint64_t t[5] = {0, 0, 0, 0, 0};
        for (size_t i = 0; i < n; i++) {
            cy = 0;
            for (int k = 0; k < 4; k++) {
                cy += (H::u128)t[k] + h_scalars_canon[4 * i + k];
                t[k] = (uint64_t)cy;
                cy >>= 64;
            }
            t[4] += (uint64_t)cy;
            // reduce mod r (r > 2^254: at most a few subtractions)
            static const uint64_t R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
            for (;;) {
                bool ge = t[4] != 0;
                if (!ge) {
                    ge = true;
                    for (int k = 3; k >= 0; k--) {
                        if (t[k] > R[k]) break;
                        if (t[k] < R[k]) {
                            ge = false;
                            break;
                        }
                    }
                }
                if (!ge) break;
                uint64_t bw = 0;
                for (int k = 0; k < 4; k++) {
                    H::u128 d = (H::u128)t[k] - R[k] - bw;
                    t[k] = (uint64_t)d;
                    bw = (uint64_t)(d >> 64) & 1;
                }
                t[4] -= bw;
            }
        }
        for (int k = 0; k < 4; k++) ksum[k] = t[k];
    }
    // all coefficients equal (d_msm with pre-scaled scalars: every coeff_i is the pack coefficient c_p;
    // d_commit / d_open sums: all ones): sum the points first, then ONE scalar multiplication
    bool same_k = n > 1;
    for (size_t i = 1; i < n && same_k; i++) same_k = std::memcmp(h_scalars_canon + 4 * i, h_scalars_canon, 32) == 0;
    auto row_work = [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; r++) {
            bool same = n > 1;
            for (size_t i = 1; i < n && same; i++) same = (pts[r * n + i].x == pts[r * n].x) && (pts[r * n + i].y == pts[r * n].y);
            if (same) {
                acc[r] = H::scalar_mul(pts[r * n], ksum);
                continue;
            }
            if (same_k && top > 0) {
                H::Jac sum = H::jac_inf();
                for (size_t i = 0; i < n; i++) sum = H::jac_add_mixed(sum, pts[r * n + i]);
                acc[r] = H::scalar_mul_jac(sum, h_scalars_canon);
                continue;
            }
            for (int b = top; b >= 0; b--) {
                acc[r] = H::jac_dbl(acc[r]);
                for (size_t i = 0; i < n; i++)
                    if ((h_scalars_canon[4 * i + b / 64] >> (b % 64)) & 1) acc[r] = H::jac_add_mixed(acc[r], pts[r * n + i]);
            }
        }
    };
    {
        const size_t nth = (top < 8 || count < 2) ? 1 : std::min<size_t>({count, (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)32});
        if (nth <= 1) {
            row_work(0, count);
        } else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < nth; t++) th.emplace_back(row_work, count * t / nth, count * (t + 1) / nth);
            for (auto& x : th) x.join();
        }
    }
    std::vector<H::Aff> outa(count);
    if (count) H::batch_to_affine(acc, outa.data());
    for (size_t r = 0; r < count; r++) {
        if (H::aff_inf(outa[r])) {
            H::write_normalised(H::jac_inf(), h_out + 18 * r);
        } else {
            std::memcpy(h_out + 18 * r, outa[r].x.data(), 48);
            std::memcpy(h_out + 18 * r + 6, outa[r].y.data(), 48);
            std::memcpy(h_out + 18 * r + 12, H::ONE.data(), 48);
        }
    }
    (void)ctx;
    return ZK_OK;
}

int g1_lincomb_host(zk_ctx* ctx, const uint64_t* h_points_jac, const uint64_t* h_scalars_canon, size_t n, uint64_t* h_out) {
    return g1_lincomb_batch_host(ctx, h_points_jac, h_scalars_canon, n, 1, h_out);
}

int dbg_g1_op(zk_ctx* ctx, int mode, const void* p, const void* q, void* h_out, size_t n) {
    if (n == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    void* d = scratch(ctx, 3, n * 192);
    if (!d) return ZK_ERR_OOM;
    hipLaunchKernelGGL(k_dbg_g1, dim3((unsigned)((n + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, p, q, d, n, mode);
    ZK_HIP(ctx, hipGetLastError());
    std::vector<uint64_t> h(n * 24);
    ZK_HIP(ctx, hipMemcpyAsync(h.data(), d, n * 192, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; i++) zkhost::write_normalised(load_xyzz_host(&h[i * 24]), (uint64_t*)h_out + 18 * i);
    return ZK_OK;
}

#else
namespace zk {
int dbg_g2_op(zk_ctx* ctx, int mode, const void* p, const void* q, void* h_out, size_t n) {
    if (n == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    void* d = scratch(ctx, 3, n * 288);
    if (!d) return ZK_ERR_OOM;
    hipLaunchKernelGGL(k_dbg_g2, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, p, q, d, n, mode);
    ZK_HIP(ctx, hipGetLastError());
    std::vector<uint64_t> h(n * 36);
    ZK_HIP(ctx, hipMemcpyAsync(h.data(), d, n * 288, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; i++) {
        zkhost::Jac2 j;
        zkhost::get_fe(j.x, &h[i * 36]);
        zkhost::get_fe(j.y, &h[i * 36 + 12]);
        zkhost::get_fe(j.z, &h[i * 36 + 24]);
        zkhost::write_normalised(j, (uint64_t*)h_out + 36 * i);
    }
    return ZK_OK;
}
#endif

}  // namespace zk
