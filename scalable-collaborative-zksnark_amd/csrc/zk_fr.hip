// zk_fr.hip -- Fr kernels of the dist-primitive hot path on gfx950:
//   K2/K3 sumcheck rounds (dsumcheck.rs:10-21, :37-85 and their c_/d_ copies),
//   K4   fold / fix_variable (mle.rs:95-103),
//   K5   open quotients q = hi - lo fused with the fold (dpoly_comm.rs:309-323),
//   K6   product tree (dacc_product.rs:31-38),
//   K8   element-wise maps and batched division (dhyperplonk.rs:233-238,251-256,326-339).
//
// Data layout in HBM: the reference's own AoS -- one Fr = 32 B (4 x u64 Montgomery limbs),
// so a lane moves one element with two 16-byte accesses and a wave reads 2 KiB contiguous.
//
// Round fusion: the reference folds the TOP variable each round (lo = tab[..m/2], hi = tab[m/2..]).
// Because every challenge is known up front, a thread that loads the 2^K elements
// {j + s*m/2^K} can run K rounds in registers: one HBM sweep per K rounds instead of per round.
// All sums are exact modular sums, so any association order is bit-identical to the reference.
#include "fp.cuh"
#include "zk_ctx.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace zk {

static constexpr int kBlock = 256;

struct ChalArgs {
    Fr c[3];
};

// ---------------------------------------------------------------------------------------
// block-level reduction of NS running sums held by every thread of a 256-thread block.
// lds: NS * 256 Fr.  Result for sum s is returned to the thread with (tid == 32*s) ... see use.
// ---------------------------------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void block_reduce_store(Fr (&acc)[NS], uint4* lds, void* dst, size_t base, size_t stride) {
    // sum s of the block is written to dst[base + s*stride].  8 groups of 32 lanes; group g owns
    // sums g, g+8, ...: 8 strided LDS loads per lane, then a 32-lane shuffle reduction.
    static_assert(NS <= 16, "two sums per 32-lane group at most");
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < NS; s++) fr_store(lds, (size_t)s * kBlock + tid, acc[s]);
    __syncthreads();
    const int grp = tid >> 5, l32 = tid & 31;
    for (int s = grp; s < NS; s += 8) {
        Fr v = fr_load(lds, (size_t)s * kBlock + l32);
#pragma unroll
        for (int i = 1; i < 8; i++) v = fr_add(v, fr_load(lds, (size_t)s * kBlock + l32 + 32 * i));
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.l[k] = __shfl_down(v.l[k], off, 32);
            v = fr_add(v, o);
        }
        if (l32 == 0) fr_store(dst, base + (size_t)s * stride, v);
    }
    __syncthreads();
}

// one sumcheck / fold / open round on the register-resident slice e[0 .. 2*half)
// MODE 0: plain sums (2), 1: product sums (3), 2: fold only, 3: open (q written by caller)
template <int MODE>
__device__ __forceinline__ void round_pair(Fr& flo, const Fr& fhi, Fr& glo, const Fr& ghi, const Fr& r, Fr* acc, Fr& q_out, bool with_t1) {
    Fr df = fr_sub(fhi, flo);
    if (MODE == 0) {
        acc[0] = fr_add(acc[0], flo);
        acc[1] = fr_add(acc[1], fhi);
    }
    if (MODE == 1) {
        Fr dg = fr_sub(ghi, glo);
        acc[0] = fr_add(acc[0], fr_mul(flo, glo));
        // t1 = sum f_hi g_hi is only computed in the very first round of a call: afterwards t0 + t1 of a round equals the
        // previous round polynomial at its challenge (dsumcheck.rs:558-588 read backwards), the host fills it in
        if (with_t1) acc[1] = fr_add(acc[1], fr_mul(fhi, ghi));
        // (2 f_hi - f_lo)(2 g_hi - g_lo) = (f_hi + df)(g_hi + dg)      dsumcheck.rs:55-72
        acc[2] = fr_add(acc[2], fr_mul(fr_add(fhi, df), fr_add(ghi, dg)));
        glo = fr_add(glo, fr_mul(r, dg));
    }
    if (MODE == 3) q_out = df;
    // lo*(1-r) + hi*r == lo + r*(hi - lo)  (same canonical field element)   dsumcheck.rs:14-19
    flo = fr_add(flo, fr_mul(r, df));
}

// ---------------------------------------------------------------------------------------
// Lazily reduced sums of the product sumcheck: t0 = sum f_lo g_lo and t2 = sum (2 f_hi - f_lo)(2 g_hi - g_lo) of a round
// are sums of PRODUCTS, so the HBM passes add the 512-bit integer products into a 544-bit integer (fp_mac_wide: the
// multiplication half of a Montgomery multiplication, no reduction half, no modular addition) and the one Montgomery
// reduction per sum happens on the host when the call ends: W0 + W1 R + W2 R^2 -> W0 R^-1 + W1 + W2 R (mod r), the same
// canonical element the reference's reduce-every-product order gives.  2^25 products of factors < 2r fit 544 bits.
// ---------------------------------------------------------------------------------------
struct Wide {
    u32 l[17];
};
static constexpr int kWideBytes = 80;  // 17 limbs + 3 words of padding: five 16-byte accesses per value
__device__ __forceinline__ void wide_add(Wide& a, const Wide& b) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 17; i++) a.l[i] = addc(a.l[i], b.l[i], c);
}
__device__ __forceinline__ Wide wide_shfl_down(const Wide& v, int off, int width) {
    Wide o;
#pragma unroll
    for (int i = 0; i < 17; i++) o.l[i] = __shfl_down(v.l[i], off, width);
    return o;
}
__device__ __forceinline__ void wide_store(void* base, size_t idx, const Wide& v) {
    uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + idx * kWideBytes);
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = make_uint4(v.l[4 * i], v.l[4 * i + 1], v.l[4 * i + 2], v.l[4 * i + 3]);
    p[4] = make_uint4(v.l[16], 0, 0, 0);
}
__device__ __forceinline__ Wide wide_load(const void* base, size_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + idx * kWideBytes);
    Wide v;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint4 x = p[i];
        v.l[4 * i] = x.x, v.l[4 * i + 1] = x.y, v.l[4 * i + 2] = x.z, v.l[4 * i + 3] = x.w;
    }
    v.l[16] = p[4].x;
    return v;
}
// a + b as 256-bit integers (both < r: no wrap), NOT reduced: a factor of a lazily reduced product
__device__ __forceinline__ Fr fr_add_nored(const Fr& a, const Fr& b) {
    Fr r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = addc(a.l[i], b.l[i], c);
    return r;
}
// the product-sumcheck round on one pair with lazily reduced sums (see round_pair<1>)
__device__ __forceinline__ void round_pair_lazy(Fr& flo, const Fr& fhi, Fr& glo, const Fr& ghi, const Fr& r, Wide& w0, Wide& w1, Wide& w2,
                                                bool with_t1) {
    const Fr df = fr_sub(fhi, flo), dg = fr_sub(ghi, glo);
    fp_mac_wide(w0.l, flo, glo);
    if (with_t1) fp_mac_wide(w1.l, fhi, ghi);
    fp_mac_wide(w2.l, fr_add_nored(fhi, df), fr_add_nored(ghi, dg));  // (2 f_hi - f_lo)(2 g_hi - g_lo), factors < 2r   dsumcheck.rs:55-72
    glo = fr_add(glo, fr_mul(r, dg));
    flo = fr_add(flo, fr_mul(r, df));
}

// Hand-over layout between the last HBM pass and the local stage behind it.  A local stage cuts its table CYCLICALLY (workgroup
// w owns the elements {w + G t}: the round pairs then stay inside the workgroup), so out of a linearly stored table every lane
// of it gathers 32 bytes from its own 128-byte line -- 10 us of the 2^20 product sumcheck's 31-us first local stage.  The pass
// that produces the table can just as well store slice w CONTIGUOUSLY (element w + G t at w S + t, S = slice length): it walks
// its outputs in tiles of 16 slices x 16 positions per workgroup (a wave: 16 x 4), so its reads are 512-byte runs, its stores
// whole 128-byte lines, and the local stage streams its slice.  Pure re-indexing of an intermediate buffer.
struct Handover {
    unsigned G = 0;      // slices (workgroups of the next stage); 0: linear table
    unsigned S = 0;      // elements per slice
};
// work item `idx` of a pass -> (table index j it computes, position it stores to)
__device__ __forceinline__ void handover_map(const Handover& h, size_t idx, size_t& j, size_t& pos) {
    if (h.G == 0) {
        j = pos = idx;
        return;
    }
    const unsigned tile = (unsigned)(idx >> 8), t = (unsigned)idx & 255u, gw = h.G >> 4;
    const unsigned w = (tile % gw) * 16 + (t & 15u), p = (tile / gw) * 16 + (t >> 4);
    j = (size_t)w + (size_t)h.G * p;
    pos = (size_t)w * h.S + p;
}

template <int MODE>
struct ModeTraits {
    static constexpr int W = (MODE == 0) ? 2 : (MODE == 1 ? 3 : 0);
    static constexpr bool TWO = (MODE == 1);
};

// ---------------------------------------------------------------------------------------
// K fused rounds over a table of length m living in HBM.  Thread handles output index j
// (grid-stride), reading f[j + s*(m>>K)], s < 2^K (each a coalesced 2-KiB wave read).
// partials layout: [(rd*W + w) * gridDim.x + blockIdx.x] Fr; product sumcheck: 544-bit integers, one per wave (see Wide)
// ---------------------------------------------------------------------------------------
// FULLT1 (product passes, test switch sc_t1_device): t1 = sum f_hi g_hi of EVERY round is accumulated on the device (3K wide
// sums instead of 2K + 1) -- the independent value the derived t1 is checked against (tests/test_gpu_bigsizes.py).
template <int K, int MODE, bool FULLT1 = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, ((K == 3 && MODE == 1) || FULLT1) ? 1 : ((K <= 2 && MODE != 1) ? 4 : 2)))) k_pass(const void* __restrict__ f, const void* __restrict__ g, void* __restrict__ fo,
                                               void* __restrict__ go, size_t m, ChalArgs ch, void* __restrict__ partials,
                                               void* __restrict__ qbase, int t1mode, Handover ho) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    constexpr int E = 1 << K;
    constexpr int NS = (W == 0) ? 1 : K * W;
    constexpr bool LAZY = (MODE == 1);      // lazily reduced sums: t0, t2 of each round + t1 of the first (FULLT1: of every round)
    constexpr int NW = LAZY ? (FULLT1 ? 3 * K : 2 * K + 1) : 1;
    extern __shared__ uint4 lds[];
    const size_t q = m >> K;
    Fr acc[LAZY ? 1 : NS];
    Wide wacc[NW];
#pragma unroll
    for (int s = 0; s < (LAZY ? 1 : NS); s++) acc[s] = fp_zero<FrCfg>();
    if (LAZY) {
#pragma unroll
        for (int a = 0; a < NW; a++)
#pragma unroll
            for (int i = 0; i < 17; i++) wacc[a].l[i] = 0;
    }

    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < q; idx += (size_t)gridDim.x * kBlock) {
        size_t j, opos;
        handover_map(ho, idx, j, opos);
        Fr ef[E], eg[TWO ? E : 1];
#pragma unroll
        for (int s = 0; s < E; s++) {
            ef[s] = fr_load(f, j + (size_t)s * q);
            if (TWO) eg[s] = fr_load(g, j + (size_t)s * q);
        }
        size_t qoff = 0;  // offset of this round's q vector relative to qbase
        size_t mcur = m;
#pragma unroll
        for (int rd = 0; rd < K; rd++) {
            const int half = E >> (rd + 1);
#pragma unroll
            for (int s = 0; s < E / 2; s++) {
                if (s < half) {
                    Fr qv;
                    if (LAZY && FULLT1)
                        round_pair_lazy(ef[s], ef[s + half], eg[TWO ? s : 0], eg[TWO ? s + half : 0], ch.c[rd], wacc[FULLT1 ? 3 * rd : 0],
                                        wacc[FULLT1 ? 3 * rd + 1 : 0], wacc[FULLT1 ? 3 * rd + 2 : 0], true);
                    else if (LAZY)
                        round_pair_lazy(ef[s], ef[s + half], eg[TWO ? s : 0], eg[TWO ? s + half : 0], ch.c[rd], wacc[LAZY ? 2 * rd : 0],
                                        wacc[LAZY ? 2 * K : 0], wacc[LAZY ? 2 * rd + 1 : 0], t1mode != 0 && rd == 0);
                    else
                        round_pair<MODE>(ef[s], ef[s + half], eg[TWO ? s : 0], eg[TWO ? s + half : 0], ch.c[rd],
                                         &acc[(W == 0) ? 0 : rd * W], qv, t1mode == 2 || (t1mode == 1 && rd == 0));
                    if (MODE == 3) fr_store(qbase, qoff + j + (size_t)s * q, qv);
                }
            }
            qoff += mcur >> 1;
            mcur >>= 1;
        }
        fr_store(fo, opos, ef[0]);
        if (TWO) fr_store(go, opos, eg[0]);
    }
    if constexpr (LAZY) {
        // one 544-bit partial per wave and sum: [(round * 3 + kind) * 4 gridDim + 4 block + wave], 80-byte slots
        const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const size_t nbw = (size_t)gridDim.x * (kBlock / 64);
#pragma unroll
        for (int a = 0; a < NW; a++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wide_add(wacc[a], wide_shfl_down(wacc[a], off, 64));
            const int slot = FULLT1 ? a : ((a == 2 * K) ? 1 : (a >> 1) * 3 + ((a & 1) ? 2 : 0));
            if (lane == 0) wide_store(partials, (size_t)slot * nbw + (size_t)blockIdx.x * (kBlock / 64) + wave, wacc[a]);
        }
    } else if constexpr (W != 0) {
        block_reduce_store<NS>(acc, lds, partials, blockIdx.x, gridDim.x);
    }
}

// ---------------------------------------------------------------------------------------
// Fold-only passes (fix_variable, mle.rs:95-103) in FLAT form.  K rounds of f <- f_lo + r (f_hi - f_lo) are one linear
// map of the 2^K elements a lane loads:  out[j] = sum_S w[S] f[j + S q],  w[S] = prod_i (bit_{K-1-i}(S) ? r_i : 1 - r_i)
// (the eq-polynomial of the K challenges, 2^K products computed on the host).  No intermediate level is needed by
// anyone, so a lane adds the 2^K integer products into ONE 544-bit sum (fp_mac_wide_s: the weights are wave-uniform
// kernel arguments, i.e. SGPR operands) and reduces once: 2^K half-multiplications + one reduction per output instead
// of 2^K - 1 full multiplications -- half the issue slots per element at K = 4 -- and 30 live registers instead of 250,
// so the sweep runs at full occupancy and is bound by HBM, not by the multiplier.  Exact integer arithmetic: the
// canonical result is the one the round-by-round order gives.
// ---------------------------------------------------------------------------------------
struct FlatW {
    Fr w[16];
};
// v (9 limbs) -= 2^S r if v >= 2^S r
template <int S>
__device__ __forceinline__ void w9_csub(u32 (&v)[9]) {
    u32 d[9], bw = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 lo = i < 8 ? FrCfg::P(i) : 0u, below = i > 0 ? FrCfg::P(i - 1) : 0u;
        const u32 c = S == 0 ? lo : ((lo << S) | (below >> (32 - S)));  // limb i of r << S
        d[i] = subb(v[i], c, bw);
    }
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] = bw ? v[i] : d[i];
}
template <int K>
__global__ void __launch_bounds__(kBlock) k_fold_flat(const void* __restrict__ f, void* __restrict__ fo, size_t m, FlatW w, Handover ho) {
    constexpr int E = 1 << K;
    const size_t q = m >> K;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < q; idx += (size_t)gridDim.x * kBlock) {
        size_t j, opos;
        handover_map(ho, idx, j, opos);
        u32 acc[17];
#pragma unroll
        for (int i = 0; i < 17; i++) acc[i] = 0;
#pragma unroll
        for (int s = 0; s < E; s++) fp_mac_wide_s(acc, fr_load(f, j + (size_t)s * q), w.w[s]);
        u32 v[9];
        fp_redc_wide(v, acc);  // < (2^K r^2 / 2^256) + r < (0.4528 * 2^K + 1) r
        if (K >= 4) w9_csub<3>(v);
        if (K >= 3) w9_csub<2>(v);
        if (K >= 2) w9_csub<1>(v);
        w9_csub<0>(v);
        Fr o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = v[i];
        fr_store(fo, opos, o);
    }
}

// The plain sumcheck's passes in the same flat form.  Its round sums t0 = sum of the low half, t1 = sum of the high half of
// every level are LINEAR in the table: with A_S = sum_j f[j + S q] (the 2^K column sums of the pass, plain additions)
// the K rounds of the pass are the sumcheck of the 2^K-element table (A_S) itself -- 2^K field elements, run on the host.
// So a lane keeps 2^K lazily reduced 288-bit column sums (one 9-limb integer addition per element) next to the wide
// sum of the fold output: one multiply-accumulate + one addition per element instead of 7/8 multiplication + 3.5
// modular additions, and no dependency between the rounds.
static constexpr int kW9Bytes = 48;  // 9 limbs + 3 words of padding
template <int K>
__global__ void __launch_bounds__(kBlock) k_plain_flat(const void* __restrict__ f, void* __restrict__ fo, size_t m, FlatW w,
                                                      void* __restrict__ partials, Handover ho) {
    constexpr int E = 1 << K;
    const size_t q = m >> K;
    u32 cs[E][9];
#pragma unroll
    for (int s = 0; s < E; s++)
#pragma unroll
        for (int i = 0; i < 9; i++) cs[s][i] = 0;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < q; idx += (size_t)gridDim.x * kBlock) {
        size_t j, opos;
        handover_map(ho, idx, j, opos);
        u32 acc[17];
#pragma unroll
        for (int i = 0; i < 17; i++) acc[i] = 0;
#pragma unroll
        for (int s = 0; s < E; s++) {
            const Fr e = fr_load(f, j + (size_t)s * q);
            fp_mac_wide_s(acc, e, w.w[s]);
            u32 c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) cs[s][i] = addc(cs[s][i], e.l[i], c);
            cs[s][8] += c;
        }
        u32 v[9];
        fp_redc_wide(v, acc);
        if (K >= 4) w9_csub<3>(v);
        if (K >= 3) w9_csub<2>(v);
        if (K >= 2) w9_csub<1>(v);
        w9_csub<0>(v);
        Fr o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = v[i];
        fr_store(fo, opos, o);
    }
    // one 288-bit partial per wave and column: [S * 4 gridDim + 4 block + wave], 48-byte slots
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t nbw = (size_t)gridDim.x * (kBlock / 64);
#pragma unroll
    for (int s = 0; s < E; s++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            u32 c = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) cs[s][i] = addc(cs[s][i], __shfl_down(cs[s][i], off, 64), c);
        }
        if (lane == 0) {
            uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<char*>(partials) + ((size_t)s * nbw + (size_t)blockIdx.x * (kBlock / 64) + wave) * kW9Bytes);
            p[0] = make_uint4(cs[s][0], cs[s][1], cs[s][2], cs[s][3]);
            p[1] = make_uint4(cs[s][4], cs[s][5], cs[s][6], cs[s][7]);
            p[2] = make_uint4(cs[s][8], 0, 0, 0);
        }
    }
}

// the per-block partial sums of ALL passes of one call, reduced in a single launch after the last
// pass (the sums of round i are not an input of round i+1): block b -> output b of pass p
struct ReducePlan {
    static constexpr int kMax = 24;
    const void* partials[kMax];  // [nsums][nb] Fr
    unsigned nb[kMax];
    unsigned first[kMax + 1];    // first output index of pass p (prefix sums of nsums); outputs are consecutive in `out`
    unsigned obase[kMax];        // first output slot of pass p in `out` (Fr) / `wout` (wide kinds)
    unsigned char wide[kMax];    // 1: the partials (and the output) of pass p are 544-bit integers in 80-byte slots; 2: 288-bit partials in 48-byte slots, 544-bit output
    int n;
};
// the challenges of a local stage travel as kernel arguments (at most log2(kLocalMaxE) = 10 rounds)
struct TailChal {
    uint64_t c[11 * 4];
};

// ---------------------------------------------------------------------------------------
// Local stage: ALL rounds that fit on a table slice living in LDS, in one launch.
//
// A table of m = G * E elements is cut cyclically: workgroup w owns the E elements {w + G*t}.  The
// round pairs (j, j + m/2) then stay inside one workgroup for log2(E) rounds (locally they are the
// pairs (t, t + E/2)), after which every workgroup is down to one element and the table to G.
// The same kernel with G = 1 finishes a call: it is the only stage of a table of <= E elements.
//
// Inside a workgroup the work is arranged by dependency, not by round:
//   phase A  the fold chain -- the only true dependency chain of a sumcheck: ONE multiplication per
//            round (f and g fold side by side in different lanes); every level of the table is kept
//            (level k at Fr index 2E - 2E/2^k: 2E elements per table in all).  Once a round fits one
//            wave the workgroup barrier is dropped (a wave's LDS operations execute in order).
//   phase B  everything the sums need, for all rounds at once: the 3(E-1) pair products of the
//            product sumcheck are independent of each other given the levels, so they are one
//            embarrassingly parallel sweep, then one segmented reduction.
// The old single-workgroup tail did the five multiplications of a pair serially in every round: 6.5 us
// per round, 65 us for the last ten rounds of a 2^20 product sumcheck.
//
// Blocks >= G of the launch reduce the per-block partial sums of the earlier stages (`plan`).
// sums layout: [(round * W + w) * G + block]  (G = 1: the final result slots).
// ---------------------------------------------------------------------------------------
static constexpr int kLocalThreads = 1024;
static constexpr unsigned kLocalMaxE = 1024;  // one table: 2E Fr = 64 KiB; two tables use E <= 512 (+ 48 KiB of parked products)

__device__ __forceinline__ unsigned lvl_off(unsigned E, int k) { return 2 * E - ((2 * E) >> k); }

__device__ __forceinline__ Fr fr_shfl_down32(const Fr& v, int off) {
    Fr o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.l[k] = __shfl_down(v.l[k], off, 32);
    return o;
}

// one output sum of an earlier stage: block `ob` of the reduce part of the launch (blockDim.x = 1024 or 256 threads)
__device__ __forceinline__ void reduce_all_body(const ReducePlan& plan, void* __restrict__ out, void* __restrict__ wout, unsigned ob,
                                                uint4* lds) {
    int p = 0;
    while (p + 1 < plan.n && ob >= plan.first[p + 1]) p++;
    const size_t s = ob - plan.first[p], nb = plan.nb[p];
    const int tid = threadIdx.x, grp = tid >> 5, l32 = tid & 31;
    const unsigned nt = blockDim.x, ngrp = nt >> 5;
    if (plan.wide[p]) {
        Wide v;
#pragma unroll
        for (int i = 0; i < 17; i++) v.l[i] = 0;
        if (plan.wide[p] == 2) {
            for (size_t i = tid; i < nb; i += nt) {
                const uint4* q4 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(plan.partials[p]) + (s * nb + i) * kW9Bytes);
                const uint4 a = q4[0], b = q4[1];
                Wide t;
#pragma unroll
                for (int k = 0; k < 17; k++) t.l[k] = 0;
                t.l[0] = a.x, t.l[1] = a.y, t.l[2] = a.z, t.l[3] = a.w, t.l[4] = b.x, t.l[5] = b.y, t.l[6] = b.z, t.l[7] = b.w, t.l[8] = q4[2].x;
                wide_add(v, t);
            }
        } else {
            for (size_t i = tid; i < nb; i += nt) wide_add(v, wide_load(plan.partials[p], s * nb + i));
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) wide_add(v, wide_shfl_down(v, o, 32));
        if (l32 == 0) wide_store(lds, grp, v);
        __syncthreads();
        if (grp == 0) {
            if ((unsigned)l32 < ngrp) {
                v = wide_load(lds, l32);
            } else {
#pragma unroll
                for (int i = 0; i < 17; i++) v.l[i] = 0;
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) wide_add(v, wide_shfl_down(v, o, 32));
            if (l32 == 0) wide_store(wout, plan.obase[p] + s, v);
        }
        return;
    }
    Fr v = fp_zero<FrCfg>();
    for (size_t i = tid; i < nb; i += nt) v = fr_add(v, fr_load(plan.partials[p], s * nb + i));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fr_add(v, fr_shfl_down32(v, o));
    if (l32 == 0) fr_store(lds, grp, v);
    __syncthreads();
    if (grp == 0) {
        v = (unsigned)l32 < ngrp ? fr_load(lds, l32) : fp_zero<FrCfg>();
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v = fr_add(v, fr_shfl_down32(v, o));
        if (l32 == 0) fr_store(out, plan.obase[p] + s, v);
    }
}

template <int MODE>
__global__ void __launch_bounds__(kLocalThreads) k_local(const void* __restrict__ f, const void* __restrict__ g, unsigned G, unsigned E,
                                                       int elog, int rounds, TailChal chal, void* __restrict__ sums,
                                                       void* __restrict__ qbase, void* __restrict__ fo, void* __restrict__ go,
                                                       ReducePlan plan, void* __restrict__ red_out, void* __restrict__ red_wide,
                                                       int t1mode, int xcd_map, int pre, int tr, unsigned long long* __restrict__ ts) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    extern __shared__ uint4 lds[];
    if (blockIdx.x >= G) {
        reduce_all_body(plan, red_out, red_wide, blockIdx.x - G, lds);
        return;
    }
    // slice of this workgroup.  The four slices that share a 128-byte line of the table (32-byte elements) go to the
    // same XCD, so the line crosses the fabric once instead of once per XCD L2 (block b runs on XCD b % 8: observed
    // placement, a speed matter only -- any bijection is correct)
    unsigned w = blockIdx.x;
    if (xcd_map && (G & 31) == 0) {
        const unsigned x = w & 7, y = w >> 3;  // XCD, slot on it
        w = 4 * (x + 8 * (y >> 2)) + (y & 3);
    }
    const unsigned tid = threadIdx.x, nt = blockDim.x, ngrp = nt >> 5;  // 1 024 threads, or 256 (knob sc_local_threads: a workgroup that fits beside two accumulation workgroups)
    int tsn = 0;
#define ZK_TS()                                                            \
    do {                                                                   \
        if (ts && blockIdx.x == 0 && tid == 0) ts[tsn++] = wall_clock64(); \
    } while (0)
    ZK_TS();
    uint4* tf = lds;                          // 2E Fr (2 uint4 each)
    uint4* tg = lds + 4 * (size_t)E;          // 2E Fr
    uint4* park = lds + (TWO ? 8 : 4) * (size_t)E;
    // `pre`: the slice has 2E elements and the first round of the stage runs straight out of the table -- pairs (t, t + E),
    // one per lane, the folded pair goes to LDS as level 0, the products of the round's sums leave through a shuffle
    // reduction.  One round more per launch at no LDS: a 2^18 product table is ONE local stage + the final one (an HBM
    // pass less per call from 2^18 on).  r0 = rounds of the stage before LDS level 0.
    const int r0 = pre ? 1 : 0;
    const int grp = tid >> 5, l32 = tid & 31;
    // element t of this workgroup's slice: cyclic in a linear table, contiguous when the pass before stored slices (struct Handover)
    const size_t sl_base = tr ? (size_t)w * ((size_t)E << r0) : (size_t)w, sl_step = tr ? 1 : (size_t)G;
#define ZK_SLICE(t) (sl_base + sl_step * (size_t)(t))
    uint4* presc = park + (MODE == 1 ? 6 * (size_t)(E - (E >> rounds)) : 0);  // 64 Fr: group partials of the pre-round
    if (pre) {
        const Fr r = fr_load(chal.c, 0);
        // one pair per VIRTUAL lane vt (2E of them with two tables: role 0 = f and t0 / t1, role 1 = g and t2; E with one); a workgroup
        // of fewer threads walks them in steps of its size -- the 32-lane groups and their partials keep their virtual numbers
        const unsigned nvt = TWO ? 2 * E : E;
        for (unsigned vt = tid; vt < ((nvt + 31u) & ~31u); vt += nt) {
            const unsigned vgrp = vt >> 5;
            Fr pa = fp_zero<FrCfg>(), pb = fp_zero<FrCfg>();
            if (TWO) {
                const unsigned role = vt >> elog, p = vt & (E - 1);
                if (role < 2) {
                    const Fr flo = fr_load(f, ZK_SLICE(p)), fhi = fr_load(f, ZK_SLICE(p + E));
                    const Fr glo = fr_load(g, ZK_SLICE(p)), ghi = fr_load(g, ZK_SLICE(p + E));
                    if (role == 0) {
                        fr_store(tf, p, fr_add(flo, fr_mul(r, fr_sub(fhi, flo))));
                        pa = fr_mul(flo, glo);
                        if (t1mode != 0) pb = fr_mul(fhi, ghi);
                    } else {
                        const Fr df = fr_sub(fhi, flo), dg = fr_sub(ghi, glo);
                        fr_store(tg, p, fr_add(glo, fr_mul(r, dg)));
                        pa = fr_mul(fr_add(fhi, df), fr_add(ghi, dg));  // (2 f_hi - f_lo)(2 g_hi - g_lo)      dsumcheck.rs:55-72
                    }
                }
            } else if (vt < E) {
                const Fr lo = fr_load(f, ZK_SLICE(vt)), hi = fr_load(f, ZK_SLICE(vt + E));
                const Fr d = fr_sub(hi, lo);
                if (MODE == 3) fr_store(qbase, w + (size_t)G * vt, d);
                fr_store(tf, vt, fr_add(lo, fr_mul(r, d)));
                pa = lo, pb = hi;
            }
            if (W != 0) {
                const bool need_b = MODE == 0 || t1mode != 0;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    pa = fr_add(pa, fr_shfl_down32(pa, o));
                    if (need_b) pb = fr_add(pb, fr_shfl_down32(pb, o));
                }
                if (l32 == 0 && vgrp < 32) {
                    fr_store(presc, vgrp, pa);
                    if (need_b) fr_store(presc, 32 + vgrp, pb);
                }
            }
        }
    } else {
        for (unsigned t = tid; t < E; t += nt) {
            fr_store(tf, t, fr_load(f, ZK_SLICE(t)));
            if (TWO) fr_store(tg, t, fr_load(g, ZK_SLICE(t)));
        }
    }
    __syncthreads();
    ZK_TS();
    if (pre && W != 0 && (unsigned)grp + 3 >= ngrp) {  // the sums of the pre-round, on the last wave (the fold chain below runs on the first)
        const unsigned ng = E >> 5;              // groups per role
        const int ws = (int)ngrp - 1 - grp;      // 0: t0 | plain lo, 1: t1 | plain hi, 2: t2
        if (ws < W && !(MODE == 1 && ws == 1 && t1mode == 0)) {
            const unsigned src = (ws == 1) ? 32 : (ws == 2 ? ng : 0);
            Fr v = (unsigned)l32 < ng ? fr_load(presc, src + l32) : fp_zero<FrCfg>();
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v = fr_add(v, fr_shfl_down32(v, o));
            if (l32 == 0) fr_store(sums, (size_t)ws * G + w, v);
        }
    }
    // ---- phase A: the fold chain ----
    // (two rounds per barrier -- four independent multiplications per output on the lanes of a quad, DPP exchange -- was
    // built and measured: a field addition costs 0.08 us at one wave per SIMD, and the double round needs 8 of them against
    // 2 per single round: 2.0 us per double round against 2 x 1.08 us.  Not kept.)
    {
        unsigned L = E;
        size_t qoff = pre ? (size_t)G * E : 0, mcur = (size_t)G * E;
        for (int k = 0; k < rounds; k++) {
            const unsigned h = L >> 1, cur = lvl_off(E, k), nxt = lvl_off(E, k + 1);
            const unsigned items = TWO ? 2 * h : h;
            const bool solo = items <= 64;  // this round and every later one fit the first wave
            if (!solo || tid < 64) {
                const Fr r = fr_load(chal.c, r0 + k);
                for (unsigned it = tid; it < items; it += nt) {
                    const bool isg = TWO && it >= h;
                    const unsigned t = isg ? it - h : it;
                    uint4* T = isg ? tg : tf;
                    const Fr lo = fr_load(T, cur + t), hi = fr_load(T, cur + t + h);
                    const Fr d = fr_sub(hi, lo);
                    if (MODE == 3) fr_store(qbase, qoff + w + (size_t)G * t, d);  // q = hi - lo   dpoly_comm.rs:312
                    fr_store(T, nxt + t, fr_add(lo, fr_mul(r, d)));              // lo + r (hi - lo)  dsumcheck.rs:14-19
                }
            }
            if (solo) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            else __syncthreads();
            ZK_TS();
            qoff += mcur >> 1;
            mcur >>= 1;
            L = h;
        }
        __syncthreads();
    }
    ZK_TS();
    const unsigned Lf = E >> rounds;
    for (unsigned t = tid; t < Lf; t += nt) {
        fr_store(fo, w + (size_t)G * t, fr_load(tf, lvl_off(E, rounds) + t));
        if (TWO) fr_store(go, w + (size_t)G * t, fr_load(tg, lvl_off(E, rounds) + t));
    }
    if (W == 0 || rounds == 0) return;
    // ---- phase B: the sums of every round ----
    const unsigned P = E - Lf;  // pairs over all rounds: round k holds E/2^(k+1) of them, from index E - E/2^k
    if (MODE == 1) {
        for (unsigned it = tid; it < 3 * P; it += nt) {
            const unsigned ws = it / P, p = it - ws * P;
            const unsigned u = E - p;  // in (E/2^(k+1), E/2^k]
            const int k = elog - (32 - __clz(u - 1));
            const unsigned hk = E >> (k + 1), t = p - (E - (E >> k)), off = lvl_off(E, k);
            if (ws == 1 && !(t1mode == 2 || (t1mode == 1 && r0 + k == 0))) continue;  // t1 is derived on the host (see round_pair)
            Fr a, b;
            if (ws == 0) {
                a = fr_load(tf, off + t);
                b = fr_load(tg, off + t);
            } else {
                a = fr_load(tf, off + t + hk);
                b = fr_load(tg, off + t + hk);
                if (ws == 2) {  // (2 f_hi - f_lo)(2 g_hi - g_lo)      dsumcheck.rs:55-72
                    a = fr_add(a, fr_sub(a, fr_load(tf, off + t)));
                    b = fr_add(b, fr_sub(b, fr_load(tg, off + t)));
                }
            }
            fr_store(park, it, fr_mul(a, b));
        }
        __syncthreads();
    }
    ZK_TS();
    for (int vid = grp; vid < rounds * W; vid += (int)ngrp) {
        const int k = vid / W, ws = vid - k * W;
        if (MODE == 1 && ws == 1 && !(t1mode == 2 || (t1mode == 1 && r0 + k == 0))) continue;
        const unsigned cnt = E >> (k + 1);
        const uint4* src = (MODE == 1) ? park : tf;
        const unsigned base = (MODE == 1) ? ws * P + (E - (E >> k)) : lvl_off(E, k) + ws * cnt;  // mode 0: lo half | hi half of level k
        Fr v = fp_zero<FrCfg>();
        for (unsigned t = l32; t < cnt; t += 32) v = fr_add(v, fr_load(src, base + t));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v = fr_add(v, fr_shfl_down32(v, o));
        if (l32 == 0) fr_store(sums, (size_t)(r0 * W + vid) * G + w, v);
    }
    ZK_TS();
    if (ts && blockIdx.x == 0 && tid == 0) ts[31] = tsn;
#undef ZK_TS
#undef ZK_SLICE
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
static int ilog2(size_t x) {
    int l = 0;
    while (((size_t)1 << (l + 1)) <= x) l++;
    return l;
}

static size_t pass_blocks(zk_ctx* ctx, size_t m, int k, int mode) {
    const size_t q = m >> k;
    size_t blocks = (q + kBlock - 1) / kBlock;
    // product passes: the resident set only (2 workgroups per CU at 2 waves/SIMD) -- every lane then runs >= 2 iterations from
    // 2^18 outputs on and the 544-bit shuffle reduction at the end of a lane's life is paid half as often (2^20: -7 us)
    const size_t env_cu = (size_t)tuning().sc_pass_wg;
    const size_t per_cu = env_cu ? env_cu : (mode == 1 ? 2 : 4);
    const size_t maxb = std::max<size_t>((size_t)ctx->cu_count * per_cu, (q + (size_t)kBlock * 32 - 1) / ((size_t)kBlock * 32));  // <= 32 grid-stride iterations per lane
    return blocks > maxb ? maxb : blocks;
}
template <int K, int MODE>
static int launch_pass(zk_ctx* ctx, hipStream_t st, const void* f, const void* g, void* fo, void* go, size_t m, const uint64_t* chal, void* partials,
                       void* qbase, int t1mode, Handover ho) {
    constexpr int W = ModeTraits<MODE>::W;
    const size_t blocks = pass_blocks(ctx, m, K, MODE);
    ChalArgs ch;
    std::memset(&ch, 0, sizeof(ch));
    std::memcpy(&ch, chal, (size_t)K * 32);
    const size_t lds = (W != 0 && MODE != 1) ? (size_t)K * W * kBlock * 32 : 0;  // (the lazily reduced sums leave through shuffles)
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_pass<K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (MODE == 1) {
        if (t1mode == 2) {  // test switch: t1 of every round on the device
            hipLaunchKernelGGL((k_pass<K, MODE, true>), dim3((unsigned)blocks), dim3(kBlock), lds, st, f, g, fo, go, m, ch, partials,
                               qbase, t1mode, ho);
            ZK_HIP(ctx, hipGetLastError());
            return ZK_OK;
        }
    }
    hipLaunchKernelGGL((k_pass<K, MODE>), dim3((unsigned)blocks), dim3(kBlock), lds, st, f, g, fo, go, m, ch, partials,
                       qbase, t1mode, ho);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

// stage geometry knobs (A/B runs): workgroups of a local stage, rounds fused per HBM pass
static unsigned sc_local_g() { return (unsigned)tuning().sc_local_g; }
// tuning sc_ts = 1: 100 MHz timestamps of the stages of workgroup 0 of every local launch of a call, printed on stderr
static unsigned long long* g_ts = nullptr;
static int g_ts_n = 0;
static unsigned long long* sc_ts_next() {
    if (tuning().sc_ts != 1) return nullptr;
    if (!g_ts && hipMalloc((void**)&g_ts, 8 * 32 * 8) != hipSuccess) return nullptr;
    return g_ts_n < 8 ? g_ts + 32 * g_ts_n++ : nullptr;
}
static void sc_ts_print() {
    if (!g_ts_n) return;
    unsigned long long h[8 * 32];
    if (hipMemcpy(h, g_ts, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int i = 0; i < g_ts_n; i++) {
        fprintf(stderr, "[sc ts] local launch %d:", i);
        for (int j = 1; j < (int)h[32 * i + 31] && j < 31; j++) fprintf(stderr, " %.2f", (double)(h[32 * i + j] - h[32 * i + j - 1]) / 100.0);
        fprintf(stderr, " us\n");
    }
    g_ts_n = 0;
}
static int sc_xcd_map() { return (int)tuning().sc_xcd; }
static int sc_flat_k() {  // rounds per flat fold pass (0: the round-by-round passes)
    const int kf = (int)tuning().sc_kf;
    return kf < 0 ? 0 : (kf > 4 ? 4 : kf);
}
static bool sc_plain_flat() { return tuning().sc_plain_flat != 0; }
static int sc_pass_k(int mode) {
    const int kp = std::max(1, std::min(3, (int)tuning().sc_kp));  // product passes
    const int k0 = std::max(1, std::min(3, (int)tuning().sc_k0));  // single-table passes
    if (mode == 2 && sc_flat_k()) return sc_flat_k();
    return mode == 1 ? kp : k0;
}

// ---- host Fr arithmetic on Montgomery-form values (4 x u64): the product sumcheck's t1 = sum f_hi g_hi of every round
// after the first is NOT computed on the device: t0 + t1 of round k equals the round polynomial of round k-1 -- the
// parabola through (0, t0), (1, t1), (2, t2) -- at the challenge of round k-1 (the verifier's check, dsumcheck.rs:558-588,
// read as an identity: sum_j f'_j g'_j with f' = f_lo + r (f_hi - f_lo)).  Exact field arithmetic: the same bits. ----
namespace hfr {
typedef unsigned __int128 u128;
static const uint64_t RM[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const uint64_t INV = 0xfffffffeffffffffULL;  // -r^-1 mod 2^64
struct F {
    uint64_t l[4];
};
static inline bool geq_r(const uint64_t* a) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > RM[i]) return true;
        if (a[i] < RM[i]) return false;
    }
    return true;
}
static inline F add(const F& a, const F& b) {
    F r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || geq_r(r.l)) {
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 t = (u128)r.l[i] - RM[i] - (uint64_t)bw;
            r.l[i] = (uint64_t)t;
            bw = (t >> 64) & 1;
        }
    }
    return r;
}
static inline F sub(const F& a, const F& b) {
    F r;
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.l[i] - b.l[i] - (uint64_t)bw;
        r.l[i] = (uint64_t)t;
        bw = (t >> 64) & 1;
    }
    if (bw) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.l[i] + RM[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
static inline F mul(const F& a, const F& b) {  // CIOS Montgomery multiplication, R = 2^256
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * INV;
        c = ((u128)m * RM[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * RM[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    F r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_r(r.l)) {
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)r.l[i] - RM[i] - (uint64_t)bw;
            r.l[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
    }
    return r;
}
static inline F half(const F& a) {  // a / 2: (a + r) / 2 when a is odd
    uint64_t t[5] = {a.l[0], a.l[1], a.l[2], a.l[3], 0};
    if (t[0] & 1) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + RM[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
        t[4] = (uint64_t)c;
    }
    F r;
    for (int i = 0; i < 4; i++) r.l[i] = (t[i] >> 1) | (t[i + 1] << 63);
    return r;
}
static const uint64_t ONEM[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};  // R mod r
static const uint64_t R2M[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};  // R^2 mod r
static inline F red(F a) {  // a < 2^256 -> a mod r (2^256 < 3 r)
    while (geq_r(a.l)) {
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 t = (u128)a.l[i] - RM[i] - (uint64_t)bw;
            a.l[i] = (uint64_t)t;
            bw = (t >> 64) & 1;
        }
    }
    return a;
}
// 17 x u32 limbs W0 + W1 R + W2 R^2 (a sum of integer products of Montgomery forms) -> its Montgomery reduction
// W0 R^-1 + W1 + W2 R mod r, canonical
static inline F from_wide(const uint32_t* w) {
    F w0, w1, w2 = {{w[16], 0, 0, 0}}, one = {{1, 0, 0, 0}}, r2;
    for (int i = 0; i < 4; i++) {
        w0.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        w1.l[i] = (uint64_t)w[8 + 2 * i] | ((uint64_t)w[8 + 2 * i + 1] << 32);
        r2.l[i] = R2M[i];
    }
    return add(add(mul(red(w0), one), red(w1)), mul(w2, r2));
}
// 9 x u32 limbs W0 + W1 2^256 (a sum of Montgomery forms, W1 < 2^32) -> its canonical residue: 2^256 = R (mod r)
static inline F from_cols(const uint32_t* w) {
    F w0, w1 = {{w[8], 0, 0, 0}}, r2;
    for (int i = 0; i < 4; i++) {
        w0.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        r2.l[i] = R2M[i];
    }
    return add(red(w0), mul(w1, r2));
}
}  // namespace hfr

// flat fold pass: weights w[S], S < 2^K, from the K challenges (Montgomery forms; MSB of S <-> the first challenge)
static size_t flat_blocks(zk_ctx* ctx, size_t m, int K, bool plain = false) {
    const size_t q = m >> K;
    size_t blocks = (q + kBlock - 1) / kBlock;
    // fold: 8 .. 4096 workgroups per CU measured, 64 and up within noise, 8 is 5 % slower.  plain: a lane ends its life with a
    // 288-bit shuffle reduction per column, so the resident set only (3 waves per SIMD at its 149 registers)
    const size_t wg_fold = (size_t)std::max(1L, tuning().sc_flat_wg), wg_plain = (size_t)std::max(1L, tuning().sc_plain_wg);
    const size_t per_cu = plain ? wg_plain : wg_fold;
    const size_t maxb = std::max<size_t>((size_t)ctx->cu_count * per_cu, (q + (size_t)kBlock * 32 - 1) / ((size_t)kBlock * 32));
    return blocks > maxb ? maxb : blocks;
}
// partials == nullptr: fold only (k_fold_flat); else the plain sumcheck's pass (k_plain_flat, K <= 3)
static int launch_fold_flat(zk_ctx* ctx, hipStream_t st, const void* f, void* fo, size_t m, int K, const uint64_t* chal, void* partials = nullptr,
                            Handover ho = Handover()) {
    using namespace hfr;
    FlatW fw;
    std::memset(&fw, 0, sizeof(fw));
    F one;
    std::memcpy(&one, hfr::ONEM, 32);
    std::vector<F> w(1, one);
    for (int i = 0; i < K; i++) {  // w_{i+1}[2S + t] = w_i[S] * (t ? r_i : 1 - r_i)
        F r;
        std::memcpy(&r, chal + 4 * i, 32);
        std::vector<F> nx(w.size() * 2);
        for (size_t S = 0; S < w.size(); S++) {
            const F hi = mul(w[S], r);
            nx[2 * S + 1] = hi;
            nx[2 * S] = sub(w[S], hi);
        }
        w.swap(nx);
    }
    for (size_t S = 0; S < w.size(); S++) std::memcpy(&fw.w[S], &w[S], 32);
    const size_t blocks = flat_blocks(ctx, m, K, partials != nullptr);
    if (partials) {
        if (K == 3) hipLaunchKernelGGL((k_plain_flat<3>), dim3((unsigned)blocks), dim3(kBlock), 0, st, f, fo, m, fw, partials, ho);
        else if (K == 2) hipLaunchKernelGGL((k_plain_flat<2>), dim3((unsigned)blocks), dim3(kBlock), 0, st, f, fo, m, fw, partials, ho);
        else if (K == 1) hipLaunchKernelGGL((k_plain_flat<1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, f, fo, m, fw, partials, ho);
        else return fail(ctx, ZK_ERR_INVALID, "internal: flat plain pass of %d rounds", K);
        ZK_HIP(ctx, hipGetLastError());
        return ZK_OK;
    }
    if (K == 4) hipLaunchKernelGGL((k_fold_flat<4>), dim3((unsigned)blocks), dim3(kBlock), 0, st, f, fo, m, fw, ho);
    else if (K == 3) hipLaunchKernelGGL((k_fold_flat<3>), dim3((unsigned)blocks), dim3(kBlock), 0, st, f, fo, m, fw, ho);
    else if (K == 2) hipLaunchKernelGGL((k_fold_flat<2>), dim3((unsigned)blocks), dim3(kBlock), 0, st, f, fo, m, fw, ho);
    else hipLaunchKernelGGL((k_fold_flat<1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, f, fo, m, fw, ho);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}


// sums: rounds x (t0, t1, t2) Montgomery Fr on the host; t1 of round 0 is the device's, every later one is derived
static void derive_t1(uint64_t* sums, const uint64_t* chal, size_t rounds) {
    using namespace hfr;
    for (size_t rd = 1; rd < rounds; rd++) {
        F a, b, c, x, t0;
        std::memcpy(&a, sums + (rd - 1) * 12, 32);
        std::memcpy(&b, sums + (rd - 1) * 12 + 4, 32);
        std::memcpy(&c, sums + (rd - 1) * 12 + 8, 32);
        std::memcpy(&x, chal + (rd - 1) * 4, 32);
        std::memcpy(&t0, sums + rd * 12, 32);
        // p(X) = a + X (B + X A),  A = (c - 2b + a) / 2,  B = (-c + 4b - 3a) / 2        dsumcheck.rs:562-575
        const F b2 = add(b, b);
        const F A = half(add(sub(c, b2), a));
        const F B = half(sub(sub(add(b2, b2), c), add(add(a, a), a)));
        const F px = add(a, mul(x, add(B, mul(x, A))));
        const F t1 = sub(px, t0);
        std::memcpy(sums + rd * 12 + 4, &t1, 32);
    }
}

template <int MODE>
static int launch_local(zk_ctx* ctx, hipStream_t st, const void* f, const void* g, unsigned G, unsigned E, int pre, int rl, const uint64_t* chal, void* sums,
                        void* qb, void* fo, void* go, const ReducePlan* rp, void* red_out, void* red_wide, int t1mode, bool want_ts, int tr = 0) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    TailChal tc;
    std::memset(&tc, 0, sizeof(tc));
    // rl = rounds on the LDS levels; pre = 1: one more round before them, straight from the 2E-element slices
    if (rl > 10) return fail(ctx, ZK_ERR_INVALID, "internal: local rounds");
    if (rl + pre) std::memcpy(tc.c, chal, (size_t)(rl + pre) * 32);
    ReducePlan none;
    std::memset(&none, 0, sizeof(none));
    const unsigned extra = rp ? rp->first[rp->n] : 0u;
    const unsigned Lf = E >> rl;
    size_t lds = (size_t)(TWO ? 4 : 2) * E * 32 + (MODE == 1 ? (size_t)3 * (E - Lf) * 32 : 0);
    lds = std::max<size_t>(lds + 64 * 32, 32 * 32);  // + the group partials of the pre-round
    int elog = 0;
    while ((1u << elog) < E) elog++;
    // per call: the attribute belongs to the CURRENT device, and one process may hold a ctx per GPU
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_local<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned nthreads = tuning().sc_local_threads == 256 ? 256u : (unsigned)kLocalThreads;
    hipLaunchKernelGGL((k_local<MODE>), dim3(G + extra), dim3(nthreads), lds, st, f, g, G, E, elog, rl, tc, sums, qb, fo, go,
                       rp ? *rp : none, red_out, red_wide, t1mode, sc_xcd_map(), pre, tr, want_ts ? sc_ts_next() : nullptr);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

// One call of the family, in three steps so that several independent calls can share the enqueue and ONE completion
// (multilinear_batch): sc_plan (geometry, scratch sizes), sc_enqueue (every launch of the call on a given stream, into
// given scratch and result blocks -- no synchronisation), sc_collect (host: the Montgomery reductions of the lazily reduced
// sums, the derived t1, the rounds of the flat plain passes; copies into the caller's arrays).
struct ScStage {
    int kind;  // 0 = pass, 1 = local (G > 1)
    int k;     // rounds
    size_t m, blocks, part_off;
    unsigned G, E;
    int pre;
    Handover ho;  // pass: how it stores its output; local stage: how its input is stored (G == 0: linear)
};
struct ScColStage {
    size_t done, k, base;
};
struct ScCall {
    // the request
    int mode = 0;  // 0 plain sums, 1 product sums, 2 fold only, 3 open quotients
    const void* d_f = nullptr;
    const void* d_g = nullptr;
    size_t len = 0, rounds = 0;
    const uint64_t* h_chal = nullptr;
    uint64_t* h_sums = nullptr;
    uint64_t* h_last_f = nullptr;
    uint64_t* h_last_g = nullptr;
    void* d_out = nullptr;
    void* d_q = nullptr;
    // sc_plan
    std::vector<ScStage> plan;
    size_t emax = 0, part_bytes = 0, res_elems = 0, res_bytes = 0;
    size_t buf_bytes[4] = {0, 0, 0, 0};  // ping-pong tables: f (len/2, len/4), g (len/2, len/4)
    bool plain_flat = false;
    // resources, assigned by the caller between plan and enqueue
    hipStream_t st = nullptr;
    void* bufs[4] = {nullptr, nullptr, nullptr, nullptr};
    char* d_part = nullptr;
    char* d_res = nullptr;   // result block: [sums rounds*W][last_f][last_g][wide sums]; pinned host memory (device-visible) or device
    char* h_res = nullptr;   // where the host reads it (== d_res when that is pinned host memory)
    bool want_ts = false;
    // sc_enqueue -> sc_collect
    bool derive = false;
    size_t wide_rounds = 0;
    std::vector<ScColStage> col_stages;
};

template <int MODE>
static int sc_plan(zk_ctx* ctx, ScCall& c) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    const size_t len = c.len, rounds = c.rounds;
    const size_t emax = TWO ? kLocalMaxE / 2 : kLocalMaxE;  // table elements a workgroup holds in LDS
    c.emax = emax;
    int use_pre = tuning().sc_pre != 0;
    // Single-table modes: the pre-round of the first local stage (its first round straight out of the table) saves a round of the
    // HBM passes but is the dearer place for it when the passes are cheap flat sweeps: it pays when it spares a whole pass (or
    // all of them), not when the round it takes over would ride in a pass that runs anyway.  Measured on MI355X (C ABI, us, pre /
    // no pre): fold 2^20 53.9 / 50.9, 2^22 70.6 / 66.8, 2^24 162 / 157 but 2^19 40.4 / 47.3; plain 2^20 64.6 / 61.1 but 2^21
    // 73.4 / 77.6, 2^24 222 / 228; open 2^20 58.0 / 54.4, 2^21 68.7 / 70.8.  Rule: no pre-round when the passes needed without it
    // are as many as with it and the extra round lands in a fold pass (up to 4 rounds each) or in a pass of at most 2 rounds.
    if (use_pre && MODE != 1 && tuning().sc_pre == 1 && rounds == (size_t)ilog2(len)) {
        const int kmax = sc_pass_k(MODE);
        const int lg = ilog2(len), base = ilog2(emax * sc_local_g());
        const int need_pre = std::max(0, lg - (base + 1)), need_no = std::max(0, lg - base);  // rounds the passes must take
        const int cnt_pre = (need_pre + kmax - 1) / kmax, cnt_no = (need_no + kmax - 1) / kmax;
        const int last_no = need_no - (cnt_no - 1) * kmax;
        if (need_pre > 0 && cnt_no == cnt_pre && (MODE == 2 || last_no <= 2)) use_pre = 0;
    }
    const size_t local_max = emax * sc_local_g() * (use_pre ? 2 : 1);  // longest table handed to a local stage (x2: its pre-round)
    const size_t fr = 32;
    // result block: [sums rounds*W][last_f][last_g]
    // (written by the last kernel straight into pinned host memory: a few dozen 32-byte stores over the link instead of
    // a copy kernel and one more kernel boundary: the same kernel time, 1-4 us less per call)
    // product sumcheck: + one 80-byte slot per sum for the lazily reduced sums of the HBM passes (see struct Wide)
    c.res_elems = rounds * W + 2;
    c.plain_flat = MODE == 0 && sc_plain_flat();
    c.res_bytes = c.res_elems * fr + (MODE == 1 ? rounds * W * kWideBytes : (c.plain_flat ? (rounds + 1) * 4 * kWideBytes : 0));  // (plain: 8 column sums per 3 rounds)
    // ---- plan: HBM passes (K rounds fused in registers) down to local_max, one multi-workgroup local
    // stage down to <= 256 elements, one single-workgroup local stage for the rest ----
    c.plan.clear();
    size_t part_bytes = 0, mm = len, dd = 0;
    while (dd < rounds && mm > emax) {
        ScStage st{};
        st.m = mm;
        if (mm > local_max) {
            st.kind = 0;
            st.k = (int)std::min<size_t>({(size_t)sc_pass_k(MODE), rounds - dd, (size_t)(ilog2(mm) - ilog2(local_max))});
            st.blocks = c.plain_flat ? flat_blocks(ctx, mm, st.k, true) : pass_blocks(ctx, mm, st.k, MODE);
        } else {
            st.kind = 1;
            st.pre = mm > emax * sc_local_g();  // (= 2 emax G: the first round runs out of the table)
            st.E = (unsigned)emax;
            st.G = (unsigned)(mm / emax) >> st.pre;
            st.k = (int)std::min<size_t>((size_t)ilog2(emax) + st.pre, rounds - dd);
            st.blocks = st.G;
        }
        st.part_off = part_bytes;
        if (MODE == 1 && st.kind == 0) part_bytes += (size_t)st.k * W * st.blocks * (kBlock / 64) * kWideBytes;  // one 544-bit partial per wave
        else if (c.plain_flat && st.kind == 0) part_bytes += ((size_t)1 << st.k) * st.blocks * (kBlock / 64) * kW9Bytes;  // 2^k column sums, one 288-bit partial per wave
        else part_bytes += (size_t)st.k * W * st.blocks * fr;
        part_bytes = (part_bytes + 15) & ~(size_t)15;
        c.plan.push_back(st);
        mm = st.kind == 0 ? mm >> st.k : (size_t)st.G * (st.E >> (st.k - st.pre));
        dd += st.k;
    }
    if ((int)c.plan.size() > ReducePlan::kMax) return fail(ctx, ZK_ERR_INVALID, "internal: too many passes");
    // a pass followed by a multi-workgroup local stage stores that stage's slices contiguously (struct Handover)
    if (tuning().sc_handover)
        for (size_t i = 0; i + 1 < c.plan.size(); i++) {
            ScStage &a = c.plan[i], &b = c.plan[i + 1];
            const unsigned S = b.E << b.pre;
            if (a.kind == 0 && b.kind == 1 && b.G >= 16 && (b.G & 15) == 0 && (S & 15) == 0 && (size_t)b.G * S == b.m) {
                a.ho.G = b.ho.G = b.G;
                a.ho.S = b.ho.S = S;
            }
        }
    c.part_bytes = W != 0 ? part_bytes : 0;
    for (size_t& b : c.buf_bytes) b = 0;
    if (!c.plan.empty()) {
        c.buf_bytes[0] = (len / 2) * fr;
        c.buf_bytes[1] = std::max<size_t>(len / 4, 1) * fr;
        if (TWO) c.buf_bytes[2] = c.buf_bytes[0], c.buf_bytes[3] = c.buf_bytes[1];
    }
    return ZK_OK;
}

template <int MODE>
static int sc_enqueue(zk_ctx* ctx, ScCall& c) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    const size_t fr = 32, len = c.len, rounds = c.rounds, emax = c.emax;
    const uint64_t* h_chal = c.h_chal;
    void* const d_out = c.d_out;
    void* const d_q = c.d_q;
    void** bufs = c.bufs;
    char* d_res = c.d_res;
    char* d_part = c.d_part;
    hipStream_t st_ = c.st;
    void* d_last_f = d_res ? d_res + rounds * W * fr : nullptr;
    void* d_last_g = d_res ? d_res + (rounds * W + 1) * fr : nullptr;
    char* d_wide = d_res ? d_res + c.res_elems * fr : nullptr;
    const std::vector<ScStage>& plan = c.plan;
    const void* cf = c.d_f;
    const void* cg = c.d_g;
    size_t m = len, done = 0;
    int flip = 0;
    // product sumcheck on tables from 2^18: t1 of every round after the first is derived on the host (derive_t1); below
    // that size the host arithmetic (~0.2 us per round) costs more than the multiplications it saves
    // (the passes keep t1 for their very first round only: a call with passes always derives)
    // (test switch sc_t1_device: never derive -- every t1 is the device's own sum, through the FULLT1 passes)
    const bool derive = MODE == 1 && !tuning().sc_t1_device && (len >= ((size_t)1 << 18) || (!plan.empty() && plan[0].kind == 0));
    c.derive = derive;
    c.wide_rounds = 0;  // rounds whose sums come back as 544-bit integers
    c.col_stages.clear();  // plain sumcheck: passes whose rounds are derived from 2^k column sums
    size_t ncols = 0;
    ReducePlan rp;
    std::memset(&rp, 0, sizeof(rp));
    const bool time_it = c.want_ts && tuning().sc_ts == 3;  // HIP events around the first stage and around the whole chain
    if (time_it) hipEventRecord(ctx->ev[0], st_);
    for (const ScStage& st : plan) {
        const int k = st.k;
        const bool final_out = (MODE == 2) && (done + k == rounds);
        void* fo = final_out ? d_out : bufs[flip];
        void* go = TWO ? bufs[2 + flip] : nullptr;
        void* qb = (MODE == 3) ? (char*)d_q + (len - m) * fr : nullptr;
        void* part = d_part ? d_part + st.part_off : nullptr;
        int rc;
        const int t1mode = !derive ? 2 : (done == 0 ? 1 : 0);  // t1 on the device: 2 every round, 1 the stage's first round only, 0 never
        if (st.kind == 1) rc = launch_local<MODE>(ctx, st_, cf, cg, st.G, st.E, st.pre, k - st.pre, h_chal + 4 * done, part, qb, fo, go, nullptr, nullptr, nullptr, t1mode, c.want_ts, st.ho.G != 0);
        else if (MODE == 2 && sc_flat_k()) rc = launch_fold_flat(ctx, st_, cf, fo, m, k, h_chal + 4 * done, nullptr, st.ho);
        else if (c.plain_flat) rc = launch_fold_flat(ctx, st_, cf, fo, m, k, h_chal + 4 * done, part, st.ho);
        else if (k == 3) rc = launch_pass<3, MODE>(ctx, st_, cf, cg, fo, go, m, h_chal + 4 * done, part, qb, t1mode, st.ho);
        else if (k == 2) rc = launch_pass<2, MODE>(ctx, st_, cf, cg, fo, go, m, h_chal + 4 * done, part, qb, t1mode, st.ho);
        else rc = launch_pass<1, MODE>(ctx, st_, cf, cg, fo, go, m, h_chal + 4 * done, part, qb, t1mode, st.ho);
        if (rc) return rc;
        if (W != 0) {  // outputs of this stage: sums of rounds done .. done+k-1, consecutive in d_res
            const bool wide = MODE == 1 && st.kind == 0;
            const bool cols = c.plain_flat && st.kind == 0;  // outputs: the 2^k column sums of the pass
            const unsigned nout = cols ? 1u << k : (unsigned)(k * W);
            rp.partials[rp.n] = part;
            rp.nb[rp.n] = (unsigned)((wide || cols) ? st.blocks * (kBlock / 64) : st.blocks);
            rp.wide[rp.n] = wide ? 1 : (cols ? 2 : 0);
            rp.obase[rp.n] = cols ? (unsigned)ncols : (unsigned)(done * W);
            if (wide) c.wide_rounds = done + k;
            if (cols) {
                c.col_stages.push_back({done, (size_t)k, ncols});
                ncols += nout;
            }
            rp.first[rp.n + 1] = rp.first[rp.n] + nout;
            rp.n++;
        }
        if (time_it && done == 0) hipEventRecord(ctx->ev[1], st_);
        cf = fo;
        cg = go;
        m = st.kind == 0 ? m >> k : (size_t)st.G * (st.E >> (k - st.pre));
        done += k;
        flip ^= 1;
    }
    if (time_it && plan.empty()) hipEventRecord(ctx->ev[1], st_);
    if (done < rounds || MODE != 2) {
        // last stage: the remaining rounds (possibly zero) in one workgroup, which also emits the final
        // table; the other blocks of the launch reduce the partial sums of the earlier stages
        const int rl = (int)(rounds - done);
        if (m > emax) return fail(ctx, ZK_ERR_INVALID, "internal: last stage too large");
        void* fo = (MODE == 2) ? d_out : d_last_f;
        void* qb = (MODE == 3) ? (char*)d_q + (len - m) * fr : nullptr;
        int rc = launch_local<MODE>(ctx, st_, cf, cg, 1u, (unsigned)m, 0, rl, h_chal + 4 * done, (void*)(d_res + done * W * fr), qb, fo, d_last_g,
                                    (W != 0 && rp.n) ? &rp : nullptr, (void*)d_res, (void*)d_wide, !derive ? 2 : (done == 0 ? 1 : 0), c.want_ts);
        if (rc) return rc;
    } else if (rounds == 0) {
        ZK_HIP(ctx, hipMemcpyAsync(d_out, c.d_f, len * fr, hipMemcpyDeviceToDevice, st_));
    }
    if (time_it) hipEventRecord(ctx->ev[2], st_);
    return ZK_OK;
}

// host side of a finished call (MODE != 2): c.h_res holds the result block
template <int MODE>
static void sc_collect(ScCall& c) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    const size_t fr = 32, rounds = c.rounds;
    char* h = c.h_res;
    if (MODE == 1) {  // the one Montgomery reduction of every lazily reduced sum
        const char* hw = h + c.res_elems * fr;
        for (size_t rd = 0; rd < c.wide_rounds; rd++)
            for (int ws = 0; ws < 3; ws++) {
                if (ws == 1 && rd != 0 && c.derive) continue;  // (derived below)
                const hfr::F v = hfr::from_wide((const uint32_t*)(hw + (rd * 3 + ws) * kWideBytes));
                std::memcpy(h + (rd * 3 + ws) * fr, &v, fr);
            }
    }
    if (MODE == 0) {  // the rounds of a flat pass = the plain sumcheck of its 2^k column sums
        const char* hw = h + c.res_elems * fr;
        for (const ScColStage& cs : c.col_stages) {
            std::vector<hfr::F> col((size_t)1 << cs.k);
            for (size_t S = 0; S < col.size(); S++) col[S] = hfr::from_cols((const uint32_t*)(hw + (cs.base + S) * kWideBytes));
            for (size_t i = 0; i < cs.k; i++) {
                const size_t half = col.size() >> 1;
                hfr::F t0 = col[0], t1 = col[half], r;
                for (size_t S = 1; S < half; S++) t0 = hfr::add(t0, col[S]), t1 = hfr::add(t1, col[half + S]);
                std::memcpy(h + ((cs.done + i) * 2) * fr, &t0, fr);
                std::memcpy(h + ((cs.done + i) * 2 + 1) * fr, &t1, fr);
                std::memcpy(&r, c.h_chal + 4 * (cs.done + i), fr);
                for (size_t S = 0; S < half; S++) col[S] = hfr::add(col[S], hfr::mul(r, hfr::sub(col[half + S], col[S])));  // dsumcheck.rs:14-19
                col.resize(half);
            }
        }
    }
    if (c.derive) derive_t1((uint64_t*)h, c.h_chal, rounds);
    if (c.h_sums && rounds * W) std::memcpy(c.h_sums, h, rounds * W * fr);
    if (c.h_last_f) std::memcpy(c.h_last_f, h + rounds * W * fr, fr);
    if (TWO && c.h_last_g) std::memcpy(c.h_last_g, h + (rounds * W + 1) * fr, fr);
}

#define ZK_SC_DISPATCH(fn, mode, ...)                  \
    ((mode) == 0 ? fn<0>(__VA_ARGS__) : (mode) == 1 ? fn<1>(__VA_ARGS__) : (mode) == 2 ? fn<2>(__VA_ARGS__) : fn<3>(__VA_ARGS__))

static int sc_validate(zk_ctx* ctx, const ScCall& c) {
    if (c.mode < 0 || c.mode > 3) return fail(ctx, ZK_ERR_INVALID, "bad mode");
    if (c.len == 0 || (c.len & (c.len - 1))) return fail(ctx, ZK_ERR_INVALID, "table length %zu is not a power of two", c.len);
    if (c.rounds > (size_t)ilog2(c.len)) return fail(ctx, ZK_ERR_INVALID, "more rounds than variables");
    return ZK_OK;
}

int multilinear_run(zk_ctx* ctx, int mode, const void* d_f, const void* d_g, size_t len, const uint64_t* h_chal, size_t rounds,
                    uint64_t* h_sums, uint64_t* h_last_f, uint64_t* h_last_g, void* d_out, void* d_q) {
    ScCall c;
    c.mode = mode, c.d_f = d_f, c.d_g = d_g, c.len = len, c.rounds = rounds, c.h_chal = h_chal;
    c.h_sums = h_sums, c.h_last_f = h_last_f, c.h_last_g = h_last_g, c.d_out = d_out, c.d_q = d_q;
    int rc = sc_validate(ctx, c);
    if (rc) return rc;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const bool host_ts = tuning().sc_ts == 2;  // host-side phases of a call on stderr
    const auto hts0 = std::chrono::steady_clock::now();
    double hts_us[6];
    int hts_n = 0;
    auto hts = [&](bool last) {
        if (!host_ts) return;
        hts_us[hts_n++] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hts0).count();
        if (last) fprintf(stderr, "[sc host] launched %.2f synced %.2f collected %.2f us\n", hts_us[0], hts_us[1], hts_us[2]);
    };
    rc = ZK_SC_DISPATCH(sc_plan, mode, ctx, c);
    if (rc) return rc;
    const bool pinned_out = tuning().sc_pinned_out != 0;
    c.d_res = (mode != 2 && pinned_out) ? (char*)pinned(ctx, c.res_bytes) : (char*)scratch(ctx, 5, c.res_bytes);
    if (!c.d_res) return ZK_ERR_OOM;
    for (int i = 0; i < 4; i++)
        if (c.buf_bytes[i]) {
            c.bufs[i] = scratch(ctx, i, c.buf_bytes[i]);
            if (!c.bufs[i]) return ZK_ERR_OOM;
        }
    if (c.part_bytes) {
        c.d_part = (char*)scratch(ctx, 4, c.part_bytes);
        if (!c.d_part) return ZK_ERR_OOM;
    }
    c.st = ctx->stream;
    c.want_ts = true;
    rc = ZK_SC_DISPATCH(sc_enqueue, mode, ctx, c);
    if (rc) return rc;
    if (mode != 2) {
        c.h_res = c.d_res;
        if (!pinned_out) {
            c.h_res = (char*)pinned(ctx, c.res_bytes);
            if (!c.h_res) return ZK_ERR_OOM;
            ZK_HIP(ctx, hipMemcpyAsync(c.h_res, c.d_res, c.res_bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
        hts(false);
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hts(false);
        sc_ts_print();
        if (tuning().sc_ts == 3) {
            hipEventElapsedTime(&ctx->sc_ms[0], ctx->ev[0], ctx->ev[1]);  // first stage (the first HBM pass of a large table)
            hipEventElapsedTime(&ctx->sc_ms[1], ctx->ev[0], ctx->ev[2]);  // every launch of the call
        }
        switch (mode) {
            case 0: sc_collect<0>(c); break;
            case 1: sc_collect<1>(c); break;
            default: sc_collect<3>(c); break;
        }
        hts(true);
    }
    return ZK_OK;
}

// Several INDEPENDENT calls of the family in one go (the ~180 sumcheck / fold / open calls of a proof come in groups that do not
// depend on each other: three per layer of the wiring identity, hyperplonk/src/dhyperplonk.rs:417-478; the six gate sumchecks
// :223-260; the opens of :383-407).  Every item gets its own slice of the scratch arenas and of the pinned result block, the
// items are spread over the ctx's auxiliary streams (a small item is a chain of 1-3 latency-bound launches: chains of different
// items overlap), and the host waits ONCE.  Results are bit-identical to the one-call-at-a-time form (same kernels, same plan).
int multilinear_batch(zk_ctx* ctx, const zk_sc_item* items, size_t count) {
    if (count == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<ScCall> calls(count);
    for (size_t i = 0; i < count; i++) {
        const zk_sc_item& it = items[i];
        ScCall& c = calls[i];
        if (it.mode < 0 || it.mode > 3) return fail(ctx, ZK_ERR_INVALID, "batch item %zu: bad mode", i);
        if (it.len == 0 || (it.len & (it.len - 1))) return fail(ctx, ZK_ERR_INVALID, "batch item %zu: table length %zu is not a power of two", i, it.len);
        const size_t n = (size_t)ilog2(it.len);
        c.mode = it.mode, c.d_f = it.d_f, c.d_g = it.d_g, c.len = it.len, c.h_chal = it.h_chal;
        c.rounds = it.mode == 2 ? std::min(n, it.n_points) : n;  // min(n, points_cnt), mle.rs:94
        c.h_sums = it.h_sums, c.h_last_f = it.h_last_f, c.h_last_g = it.h_last_g;
        c.d_out = it.mode == 2 ? it.d_out : nullptr;
        c.d_q = it.mode == 3 ? it.d_out : nullptr;
        const bool ok = it.d_f && (it.mode != 1 || it.d_g) && (c.rounds == 0 || it.h_chal) && (it.mode == 2 || it.h_last_f) && (it.mode != 1 || it.h_last_g) &&
                        ((it.mode != 0 && it.mode != 1) || n == 0 || it.h_sums) && ((it.mode != 2 && !(it.mode == 3 && it.len > 1)) || it.d_out);
        if (!ok) return fail(ctx, ZK_ERR_INVALID, "batch item %zu: null argument", i);
    }
    size_t need[5] = {0, 0, 0, 0, 0}, res_total = 0;
    std::vector<size_t> off(count * 5), res_off(count);
    for (size_t i = 0; i < count; i++) {
        ScCall& c = calls[i];
        int rc = sc_validate(ctx, c);
        if (!rc) rc = ZK_SC_DISPATCH(sc_plan, c.mode, ctx, c);
        if (rc) return rc;
        for (int b = 0; b < 4; b++) {
            off[i * 5 + b] = need[b];
            need[b] += (c.buf_bytes[b] + 255) & ~(size_t)255;
        }
        off[i * 5 + 4] = need[4];
        need[4] += (c.part_bytes + 255) & ~(size_t)255;
        res_off[i] = res_total;
        res_total += (c.res_bytes + 63) & ~(size_t)63;
    }
    char* base[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < 5; b++)
        if (need[b]) {
            base[b] = (char*)scratch(ctx, b, need[b]);
            if (!base[b]) return ZK_ERR_OOM;
        }
    char* hres = (char*)pinned(ctx, std::max<size_t>(res_total, 64));
    if (!hres) return ZK_ERR_OOM;
    // (knob sc_pinned_out = 0, as in the single calls: results land in device memory and are copied to the host afterwards)
    const bool pinned_out = tuning().sc_pinned_out != 0;
    char* dres = pinned_out ? hres : (char*)scratch(ctx, 5, std::max<size_t>(res_total, 64));
    if (!dres) return ZK_ERR_OOM;
    zk_ctx::MsmLane& L = ctx->lanes[0];
    const int nst = count > 1 ? zk_ctx::kAux : 0;  // streams besides the ctx stream
    if (nst) {
        hipEventRecord(L.ev_fork, ctx->stream);
        for (int k = 0; k < nst; k++) hipStreamWaitEvent(L.aux[k], L.ev_fork, 0);
    }
    // largest items first, round-robin over the streams: the long chains start early, the short ones fill in beside them
    std::vector<size_t> order(count);
    for (size_t i = 0; i < count; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return calls[a].len > calls[b].len; });
    int rc = ZK_OK;
    for (size_t oi = 0; oi < count && !rc; oi++) {
        const size_t i = order[oi];
        ScCall& c = calls[i];
        for (int b = 0; b < 4; b++) c.bufs[b] = c.buf_bytes[b] ? base[b] + off[i * 5 + b] : nullptr;
        c.d_part = c.part_bytes ? base[4] + off[i * 5 + 4] : nullptr;
        c.d_res = dres + res_off[i];
        c.h_res = hres + res_off[i];
        c.st = nst ? (oi % (size_t)(nst + 1) == 0 ? ctx->stream : L.aux[oi % (size_t)(nst + 1) - 1]) : ctx->stream;
        c.want_ts = false;
        rc = ZK_SC_DISPATCH(sc_enqueue, c.mode, ctx, c);
    }
    if (nst) {  // join (also on the error path: nothing may still be running on the auxiliary streams when we return)
        for (int k = 0; k < nst; k++) {
            hipEventRecord(L.ev_join[k], L.aux[k]);
            hipStreamWaitEvent(ctx->stream, L.ev_join[k], 0);
        }
    }
    if (!pinned_out && !rc && res_total) hipMemcpyAsync(hres, dres, res_total, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamSynchronize(batch)");
    for (size_t i = 0; i < count; i++) {
        ScCall& c = calls[i];
        switch (c.mode) {
            case 0: sc_collect<0>(c); break;
            case 1: sc_collect<1>(c); break;
            case 3: sc_collect<3>(c); break;
            default: break;
        }
    }
    return ZK_OK;
}

// ---------------------------------------------------------------------------------------
// K6 product tree.  Level l >= 1 lives at tree[2N - 2N/2^l ...) with N/2^l elements and
// element i of level l+1 = level_l[2i] * level_l[2i+1]  (sub_index, dacc_product.rs:18-23).
// A 256-thread block owns 512 consecutive inputs of some level and produces up to 9 levels,
// exchanging intermediate products through LDS; each level's outputs of a block are contiguous.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_tree(const void* __restrict__ x, void* __restrict__ tree, size_t N, int in_level,
                                               size_t in_len, int levels, int copy_leaves) {
    __shared__ uint4 buf[2 * kBlock];  // 256 Fr
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * 2 * kBlock;  // first input element of this block
    const size_t i0 = base + 2 * (size_t)tid;
    const size_t in_off = (in_level == 0) ? 0 : 2 * N - ((2 * N) >> in_level);
    const void* src = (in_level == 0 && copy_leaves) ? x : tree;
    Fr v = fp_zero<FrCfg>();
    const bool active = i0 + 1 < in_len;
    if (active) {
        Fr a = fr_load(src, (in_level == 0 && copy_leaves) ? i0 : in_off + i0);
        Fr b = fr_load(src, (in_level == 0 && copy_leaves) ? i0 + 1 : in_off + i0 + 1);
        if (copy_leaves) {
            fr_store(tree, i0, a);
            fr_store(tree, i0 + 1, b);
        }
        v = fr_mul(a, b);
    }
    // level in_level+1: this block's outputs are [base/2, base/2 + 256)
    int lvl = in_level + 1;
    size_t cnt = kBlock;  // number of outputs this block may hold at the current level
    size_t lvl_len = in_len >> 1;
    for (int step = 0; step < levels; step++) {
        const size_t off = 2 * N - ((2 * N) >> lvl);
        const size_t blk_first = (base >> (step + 1));
        if ((size_t)tid < cnt && blk_first + tid < lvl_len) fr_store(tree, off + blk_first + tid, v);
        if (step + 1 == levels) break;
        fr_store(buf, tid, v);
        __syncthreads();
        cnt >>= 1;
        lvl_len >>= 1;
        lvl++;
        if ((size_t)tid < cnt && (blk_first >> 1) + tid < lvl_len) v = fr_mul(fr_load(buf, 2 * tid), fr_load(buf, 2 * tid + 1));
        __syncthreads();
    }
}

__global__ void k_tree_finish(void* tree, size_t N) {
    if (threadIdx.x == 0) fr_store(tree, 2 * N - 1, fp_zero<FrCfg>());
}

int product_tree(zk_ctx* ctx, const void* d_x, size_t N, void* d_tree) {
    if (N == 0 || (N & (N - 1))) return fail(ctx, ZK_ERR_INVALID, "product tree size %zu is not a power of two", N);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    if (N == 1) {
        // tree = x || x, then tree[1] = 0   (dacc_product.rs:32-38 with an empty loop)
        ZK_HIP(ctx, hipMemcpyAsync(d_tree, d_x, 32, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_tree_finish, dim3(1), dim3(64), 0, ctx->stream, d_tree, N);
        return ZK_OK;
    }
    const int total_levels = ilog2(N);  // levels 1..log2 N
    int in_level = 0;
    size_t in_len = N;
    while (in_level < total_levels) {
        int levels = std::min(9, total_levels - in_level);
        size_t blocks = (in_len + 2 * kBlock - 1) / (2 * kBlock);
        hipLaunchKernelGGL(k_tree, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d_x, d_tree, N, in_level, in_len, levels,
                           in_level == 0 ? 1 : 0);
        in_level += levels;
        in_len >>= levels;
    }
    hipLaunchKernelGGL(k_tree_finish, dim3(1), dim3(64), 0, ctx->stream, d_tree, N);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

// ---------------------------------------------------------------------------------------
// K8 element-wise maps
// ---------------------------------------------------------------------------------------
template <int OP>
__global__ void __launch_bounds__(kBlock) k_fr_binary(const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ out,
                                                    size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        Fr x = fr_load(a, i), y = fr_load(b, i);
        Fr r = (OP == 0) ? fr_add(x, y) : (OP == 1) ? fr_sub(x, y) : fr_mul(x, y);
        fr_store(out, i, r);
    }
}
__global__ void __launch_bounds__(kBlock) k_fr_axpb(const void* __restrict__ a, const void* __restrict__ b, Fr alpha, Fr beta,
                                                  void* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        Fr r = fr_add(fr_mul(alpha, fr_load(b, i)), beta);
        if (a) r = fr_add(r, fr_load(a, i));  // a == nullptr: out = alpha*b + beta
        fr_store(out, i, r);
    }
}
template <class C, int OP>
__global__ void __launch_bounds__(kBlock) k_fp_binary(const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ out,
                                                    size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        Fp<C> x = fp_load<C>(a, i), y = fp_load<C>(b, i);
        Fp<C> r = (OP == 0) ? fp_add<C>(x, y) : (OP == 1) ? fp_sub<C>(x, y) : fp_mul<C>(x, y);
        fp_store<C>(out, i, r);
    }
}

// K9 on Fr: a small PUBLIC matrix (PSS pack / unpack / unpack2 / degree-reduction maps,
// secret-sharing/src/pss.rs:93-171, degree_reduce.rs:17-23) applied to k vectors at once:
//   out[j*osv + r*osr] = sum_c M[r][c] * in[j*isv + c*isc]        (strides in elements)
// One lane per output element; the matrix (rows*cols*32 B) is read through L2.
__global__ void __launch_bounds__(kBlock) k_fr_apply_matrix(const void* __restrict__ M, size_t rows, size_t cols,
                                                          const void* __restrict__ in, size_t isv, size_t isc,
                                                          void* __restrict__ out, size_t osv, size_t osr, size_t k) {
    const size_t total = k * rows;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t r = t / k, j = t % k;  // consecutive lanes -> consecutive vectors of the same row
        Fr acc = fp_zero<FrCfg>();
        for (size_t c = 0; c < cols; c++) acc = fr_add(acc, fr_mul(fr_load(M, r * cols + c), fr_load(in, j * isv + c * isc)));
        fr_store(out, j * osv + r * osr, acc);
    }
}

// K9 on Fr by transforms -- what the reference itself runs (ark-poly radix-2 (coset) FFTs, pss.rs:93-171):
//   interpolate on a domain of size A, keep / zero-extend to the coefficients the next domain holds, evaluate on
//   a domain of size B.  pack_from_public: A = 2l (coset g), B = 8l;  unpack: A = 8l, B = 2l (coset);  unpack2:
//   A = 8l, B = 4l (coset), every second slot.  The coset factors and 1/A are folded into ONE scale per
//   coefficient (scale[k] = A^-1 (offB / offA)^k).  Cost per vector A/2 log A + B/2 log B + min(A, B)
//   multiplications instead of rows x cols of the dense map: 3.5x less at l = 8, 7x at l = 16.
// A workgroup holds 512 / max(A, B) vectors in LDS, max(A, B) / 2 lanes per vector; both transforms are
// decimation-in-time on bit-reversed input, so the resize happens on naturally ordered coefficients.
struct NttMapArgs {
    const void* winv;   // A/2 powers of omega_A^-1 (Montgomery)
    const void* w;      // B/2 powers of omega_B
    const void* scale;  // min(A, B) coefficients' scale factors
    unsigned A, B, logA, logB, nin, take, step;
};
__device__ __forceinline__ unsigned bitrev(unsigned x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }
__device__ __forceinline__ void ntt_dit(uint4* x, unsigned N, unsigned logN, const void* __restrict__ tw, unsigned lane, unsigned lanes) {
    for (unsigned s = 1; s <= logN; s++) {
        const unsigned half = 1u << (s - 1);
        for (unsigned j = lane; j < N / 2; j += lanes) {
            const unsigned grp = j >> (s - 1), pos = j & (half - 1);
            const unsigned i0 = (grp << s) + pos, i1 = i0 + half;
            const Fr t = fr_mul(fr_load(tw, (size_t)pos << (logN - s)), fr_load(x, i1));
            const Fr a = fr_load(x, i0);
            fr_store(x, i0, fr_add(a, t));
            fr_store(x, i1, fr_sub(a, t));
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(kBlock) k_fr_ntt_map(NttMapArgs a, const void* __restrict__ in, size_t isv, size_t isc,
                                                     void* __restrict__ out, size_t osv, size_t osr, size_t k) {
    extern __shared__ uint4 lds[];
    const unsigned M = a.A > a.B ? a.A : a.B;
    const unsigned lanes = M / 2 ? M / 2 : 1, vpb = kBlock / lanes;  // vectors per block
    const unsigned v = threadIdx.x / lanes, lane = threadIdx.x % lanes;
    const size_t j = (size_t)blockIdx.x * vpb + v;
    const bool live = v < vpb && j < k;  // (dead lanes still walk the barriers)
    uint4* x = lds + (size_t)v * 2 * M * 2;  // two buffers of M Fr per vector
    uint4* y = x + 2 * M;
    if (v < vpb) {
        for (unsigned i = lane; i < a.A; i += lanes)
            fr_store(x, bitrev(i, a.logA), (live && i < a.nin) ? fr_load(in, j * isv + (size_t)i * isc) : fp_zero<FrCfg>());
    }
    __syncthreads();
    if (v < vpb) ntt_dit(x, a.A, a.logA, a.winv, lane, lanes);
    else
        for (unsigned s = 1; s <= a.logA; s++) __syncthreads();
    if (v < vpb) {
        const unsigned keep = a.A < a.B ? a.A : a.B;
        for (unsigned i = lane; i < a.B; i += lanes)
            fr_store(y, bitrev(i, a.logB), i < keep ? fr_mul(fr_load(a.scale, i), fr_load(x, i)) : fp_zero<FrCfg>());
    }
    __syncthreads();
    if (v < vpb) ntt_dit(y, a.B, a.logB, a.w, lane, lanes);
    else
        for (unsigned s = 1; s <= a.logB; s++) __syncthreads();
    if (live)
        for (unsigned r = lane; r < a.take; r += lanes) fr_store(out, j * osv + (size_t)r * osr, fr_load(y, (size_t)r * a.step));
}

// K7 strided splits (dacc_product.rs:41-55, dhyperplonk.rs:344-359): even[i] = t[2i], odd[i] = t[2i+1]
__global__ void __launch_bounds__(kBlock) k_fr_deinterleave(const void* __restrict__ t, void* __restrict__ even,
                                                          void* __restrict__ odd, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        // one lane moves a 64-byte pair: both halves of the read are contiguous across the wave
        fr_store(even, i, fr_load(t, 2 * i));
        fr_store(odd, i, fr_load(t, 2 * i + 1));
    }
}

static unsigned grid_for(zk_ctx* ctx, size_t n) {
    size_t b = (n + kBlock - 1) / kBlock;
    size_t maxb = (size_t)ctx->cu_count * 8;
    return (unsigned)std::max<size_t>(1, std::min(b, maxb));
}

int fr_binary(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n) {
    if (n == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    unsigned g = grid_for(ctx, n);
    if (op == 0) hipLaunchKernelGGL((k_fr_binary<0>), dim3(g), dim3(kBlock), 0, ctx->stream, a, b, out, n);
    else if (op == 1) hipLaunchKernelGGL((k_fr_binary<1>), dim3(g), dim3(kBlock), 0, ctx->stream, a, b, out, n);
    else hipLaunchKernelGGL((k_fr_binary<2>), dim3(g), dim3(kBlock), 0, ctx->stream, a, b, out, n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}
int fr_apply_matrix(zk_ctx* ctx, const uint64_t* h_matrix, size_t rows, size_t cols, const void* d_in, size_t isv, size_t isc,
                    void* d_out, size_t osv, size_t osr, size_t k) {
    if (k == 0 || rows == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    void* d_m = scratch(ctx, 10, rows * cols * 32);
    if (!d_m) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipMemcpyAsync(d_m, h_matrix, rows * cols * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h_matrix is caller memory (pageable)
    hipLaunchKernelGGL(k_fr_apply_matrix, dim3(grid_for(ctx, k * rows)), dim3(kBlock), 0, ctx->stream, (const void*)d_m, rows, cols,
                       d_in, isv, isc, d_out, osv, osr, k);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}
int fr_ntt_map(zk_ctx* ctx, size_t A, const uint64_t* h_winv, size_t B, const uint64_t* h_w, const uint64_t* h_scale, size_t nin, size_t take,
               size_t step, const void* d_in, size_t isv, size_t isc, void* d_out, size_t osv, size_t osr, size_t k) {
    if (k == 0 || take == 0) return ZK_OK;
    auto pow2 = [](size_t x) { return x && !(x & (x - 1)); };
    if (!pow2(A) || !pow2(B) || A > 512 || B > 512 || nin > A || (take - 1) * step >= B)
        return fail(ctx, ZK_ERR_INVALID, "zk_fr_ntt_map: domain sizes must be powers of two <= 512 and the selection inside the output domain");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t keep = std::min(A, B), tw_elems = A / 2 + B / 2 + keep + 2;
    char* d_t = (char*)scratch(ctx, 10, tw_elems * 32);
    if (!d_t) return ZK_ERR_OOM;
    char* h = (char*)pinned(ctx, tw_elems * 32);
    if (!h) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the pinned staging area may still feed an earlier copy
    std::memcpy(h, h_winv, (A / 2) * 32);
    std::memcpy(h + (A / 2) * 32, h_w, (B / 2) * 32);
    std::memcpy(h + (A / 2 + B / 2) * 32, h_scale, keep * 32);
    ZK_HIP(ctx, hipMemcpyAsync(d_t, h, (A / 2 + B / 2 + keep) * 32, hipMemcpyHostToDevice, ctx->stream));
    NttMapArgs a;
    a.winv = d_t;
    a.w = d_t + (A / 2) * 32;
    a.scale = d_t + (A / 2 + B / 2) * 32;
    a.A = (unsigned)A;
    a.B = (unsigned)B;
    a.logA = (unsigned)ilog2(A);
    a.logB = (unsigned)ilog2(B);
    a.nin = (unsigned)nin;
    a.take = (unsigned)take;
    a.step = (unsigned)step;
    const size_t M = std::max(A, B), lanes = std::max<size_t>(M / 2, 1), vpb = kBlock / lanes;
    const size_t lds = vpb * 2 * M * 32;
    hipLaunchKernelGGL(k_fr_ntt_map, dim3((unsigned)((k + vpb - 1) / vpb)), dim3(kBlock), lds, ctx->stream, a, d_in, isv, isc, d_out, osv, osr, k);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the twiddle scratch is reused by the next call
    return ZK_OK;
}
int fr_deinterleave(zk_ctx* ctx, const void* t, void* even, void* odd, size_t n) {
    if (n == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fr_deinterleave, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, t, even, odd, n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}
int fr_axpb(zk_ctx* ctx, const void* a, const void* b, const uint64_t* alpha, const uint64_t* beta, void* out, size_t n) {
    if (n == 0) return ZK_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    Fr al, be;
    std::memcpy(&al, alpha, 32);
    std::memcpy(&be, beta, 32);
    hipLaunchKernelGGL(k_fr_axpb, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, a, b, al, be, out, n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

// ---------------------------------------------------------------------------------------
// batched division out = num / den (dhyperplonk.rs:339).  Montgomery's trick per thread over
// CH strided elements (coalesced: element k of thread t is t + k*T), prefix products parked in
// `out`, one Fermat inversion per thread.  Identical result to per-element inverse().
// ---------------------------------------------------------------------------------------
static constexpr int kDivChunk = 16;
__global__ void __launch_bounds__(kBlock) k_batch_div(const void* __restrict__ num, const void* __restrict__ den,
                                                    void* __restrict__ out, size_t n, size_t T, int* __restrict__ zero_flag) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
    Fr p = fp_one<FrCfg>();
    int cnt = 0;
    bool zero = false;
    for (int k = 0; k < kDivChunk; k++) {
        size_t i = t + (size_t)k * T;
        if (i >= n) break;
        Fr d = fr_load(den, i);
        zero |= fp_is_zero<FrCfg>(d);
        fr_store(out, i, p);  // prefix product BEFORE element k
        p = fr_mul(p, d);
        cnt++;
    }
    if (zero) {
        atomicOr(zero_flag, 1);
        return;
    }
    Fr inv = fp_inv<FrCfg>(p);
    for (int k = cnt - 1; k >= 0; k--) {
        size_t i = t + (size_t)k * T;
        Fr pre = fr_load(out, i);
        Fr di = fr_mul(inv, pre);  // 1/den_k
        inv = fr_mul(inv, fr_load(den, i));
        fr_store(out, i, fr_mul(fr_load(num, i), di));
    }
}

int fr_batch_div(zk_ctx* ctx, const void* num, const void* den, void* out, size_t n) {
    if (n == 0) return ZK_OK;
    if (out == den || out == num) return fail(ctx, ZK_ERR_INVALID, "zk_fr_batch_div: out must not alias an input");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    int* flag = (int*)scratch(ctx, 6, 256);
    if (!flag) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    size_t T = (n + kDivChunk - 1) / kDivChunk;
    hipLaunchKernelGGL(k_batch_div, dim3((unsigned)((T + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, num, den, out, n, T,
                       flag);
    int h = 0;
    ZK_HIP(ctx, hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h) return fail(ctx, ZK_ERR_DIV_ZERO, "zero denominator");
    return ZK_OK;
}

}  // namespace zk
