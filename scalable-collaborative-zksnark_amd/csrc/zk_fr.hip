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
#include <cstring>
#include <vector>

namespace zk {

static constexpr int kBlock = 256;
static constexpr size_t kTailMax = 2048;  // elements handled by the single-workgroup tail kernel

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
__device__ __forceinline__ void round_pair(Fr& flo, const Fr& fhi, Fr& glo, const Fr& ghi, const Fr& r, Fr* acc, Fr& q_out) {
    Fr df = fr_sub(fhi, flo);
    if (MODE == 0) {
        acc[0] = fr_add(acc[0], flo);
        acc[1] = fr_add(acc[1], fhi);
    }
    if (MODE == 1) {
        Fr dg = fr_sub(ghi, glo);
        acc[0] = fr_add(acc[0], fr_mul(flo, glo));
        acc[1] = fr_add(acc[1], fr_mul(fhi, ghi));
        // (2 f_hi - f_lo)(2 g_hi - g_lo) = (f_hi + df)(g_hi + dg)      dsumcheck.rs:55-72
        acc[2] = fr_add(acc[2], fr_mul(fr_add(fhi, df), fr_add(ghi, dg)));
        glo = fr_add(glo, fr_mul(r, dg));
    }
    if (MODE == 3) q_out = df;
    // lo*(1-r) + hi*r == lo + r*(hi - lo)  (same canonical field element)   dsumcheck.rs:14-19
    flo = fr_add(flo, fr_mul(r, df));
}

template <int MODE>
struct ModeTraits {
    static constexpr int W = (MODE == 0) ? 2 : (MODE == 1 ? 3 : 0);
    static constexpr bool TWO = (MODE == 1);
};

// ---------------------------------------------------------------------------------------
// K fused rounds over a table of length m living in HBM.  Thread handles output index j
// (grid-stride), reading f[j + s*(m>>K)], s < 2^K (each a coalesced 2-KiB wave read).
// partials layout: [(rd*W + w) * gridDim.x + blockIdx.x]
// ---------------------------------------------------------------------------------------
template <int K, int MODE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, (K == 3 && MODE == 1) ? 1 : 2))) k_pass(const void* __restrict__ f, const void* __restrict__ g, void* __restrict__ fo,
                                               void* __restrict__ go, size_t m, ChalArgs ch, void* __restrict__ partials,
                                               void* __restrict__ qbase) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    constexpr int E = 1 << K;
    constexpr int NS = (W == 0) ? 1 : K * W;
    extern __shared__ uint4 lds[];
    const size_t q = m >> K;
    Fr acc[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) acc[s] = fp_zero<FrCfg>();

    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < q; j += (size_t)gridDim.x * kBlock) {
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
            constexpr int dummy = 0;
            (void)dummy;
            const int half = E >> (rd + 1);
#pragma unroll
            for (int s = 0; s < E / 2; s++) {
                if (s < half) {
                    Fr qv;
                    round_pair<MODE>(ef[s], ef[s + half], eg[TWO ? s : 0], eg[TWO ? s + half : 0], ch.c[rd],
                                     &acc[(W == 0) ? 0 : rd * W], qv);
                    if (MODE == 3) fr_store(qbase, qoff + j + (size_t)s * q, qv);
                }
            }
            qoff += mcur >> 1;
            mcur >>= 1;
        }
        fr_store(fo, j, ef[0]);
        if (TWO) fr_store(go, j, eg[0]);
    }
    if (W != 0) block_reduce_store<NS>(acc, lds, partials, blockIdx.x, gridDim.x);
}

// the per-block partial sums of ALL passes of one call, reduced in a single launch after the last
// pass (the sums of round i are not an input of round i+1): block b -> output b of pass p
struct ReducePlan {
    static constexpr int kMax = 24;
    const void* partials[kMax];  // [nsums][nb] Fr
    unsigned nb[kMax];
    unsigned first[kMax + 1];    // first output index of pass p (prefix sums of nsums); outputs are consecutive in `out`
    int n;
};
__global__ void __launch_bounds__(kBlock) k_reduce_all(ReducePlan plan, void* __restrict__ out) {
    extern __shared__ uint4 lds[];
    int p = 0;
    while (p + 1 < plan.n && blockIdx.x >= plan.first[p + 1]) p++;
    const size_t s = blockIdx.x - plan.first[p], nb = plan.nb[p];
    Fr acc[1];
    acc[0] = fp_zero<FrCfg>();
    for (size_t i = threadIdx.x; i < nb; i += kBlock) acc[0] = fr_add(acc[0], fr_load(plan.partials[p], s * nb + i));
    block_reduce_store<1>(acc, lds, out, blockIdx.x, 0);
}

// the tail's challenges travel as kernel arguments (at most log2(kTailMax) = 11 rounds)
struct TailChal {
    uint64_t c[11 * 4];
};

// ---------------------------------------------------------------------------------------
// Tail: all remaining rounds on a table of m <= kTailMax elements inside ONE workgroup, the
// table(s) resident in LDS (2048 x 32 B x 2 tables = 128 KiB of the CU's 160 KiB).
// Writes: sums -> sums_out[(rd)*W + w]; q -> qbase; final table (m >> rounds) -> fo / go.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_tail(const void* __restrict__ f, const void* __restrict__ g, size_t m, int rounds,
                                               TailChal chal, void* __restrict__ sums_out,
                                               void* __restrict__ qbase, void* __restrict__ fo, void* __restrict__ go) {
    // One barrier per round: the per-lane partial sums of every round are parked in LDS and ALL
    // rounds are reduced together at the end (the sums of round i are not an input of round i+1).
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    constexpr int NS = (W == 0) ? 1 : W;
    extern __shared__ uint4 lds[];
    uint4* tf = lds;
    uint4* tg = lds + 2 * m;                 // 2 uint4 per Fr
    uint4* part = lds + (TWO ? 4 : 2) * m;   // parked partial sums
    const int tid = threadIdx.x;
    for (size_t i = tid; i < m; i += kBlock) {
        fr_store(tf, i, fr_load(f, i));
        if (TWO) fr_store(tg, i, fr_load(g, i));
    }
    __syncthreads();
    size_t qoff = 0, poff = 0;
    size_t mm = m;
    for (int rd = 0; rd < rounds; rd++) {
        const size_t h = mm >> 1;
        const size_t cnt = h < (size_t)kBlock ? h : (size_t)kBlock;
        const Fr r = fr_load(chal.c, rd);  // kernel-argument segment: no H2D copy, no extra buffer
        Fr acc[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) acc[s] = fp_zero<FrCfg>();
        if (MODE == 1 && 3 * h <= (size_t)kBlock) {
            // late rounds: more lanes than pairs -> three lanes per pair, two dependent multiplications
            // per lane instead of five (the round time is the latency of that chain)
            const size_t role = (size_t)tid / h, j = (size_t)tid % h;
            Fr nf, ng;
            bool wf = false, wg = false;
            if (role < 3) {
                Fr flo = fr_load(tf, j), fhi = fr_load(tf, j + h), glo = fr_load(tg, j), ghi = fr_load(tg, j + h);
                Fr df = fr_sub(fhi, flo), dg = fr_sub(ghi, glo);
                if (role == 0) {
                    acc[0] = fr_mul(flo, glo);
                    nf = fr_add(flo, fr_mul(r, df));
                    wf = true;
                } else if (role == 1) {
                    acc[0] = fr_mul(fhi, ghi);
                    ng = fr_add(glo, fr_mul(r, dg));
                    wg = true;
                } else {
                    acc[0] = fr_mul(fr_add(fhi, df), fr_add(ghi, dg));
                }
            }
            __syncthreads();  // every lane has read its operands before anyone overwrites the tables
            if (wf) fr_store(tf, j, nf);
            if (wg) fr_store(tg, j, ng);
            if (role < 3) fr_store(part, poff + role * cnt + j, acc[0]);  // cnt == h here
            __syncthreads();
        } else {
            for (size_t j = tid; j < h; j += kBlock) {
                Fr flo = fr_load(tf, j), fhi = fr_load(tf, j + h), glo, ghi, qv;
                if (TWO) {
                    glo = fr_load(tg, j);
                    ghi = fr_load(tg, j + h);
                }
                round_pair<MODE>(flo, fhi, glo, ghi, r, acc, qv);
                if (MODE == 3) fr_store(qbase, qoff + j, qv);
                fr_store(tf, j, flo);
                if (TWO) fr_store(tg, j, glo);
            }
            if (W != 0 && (size_t)tid < cnt) {
#pragma unroll
                for (int w = 0; w < NS; w++) fr_store(part, poff + (size_t)w * cnt + tid, acc[w]);
            }
            __syncthreads();
        }
        qoff += h;
        poff += cnt * W;
        mm = h;
    }
    for (size_t i = tid; i < mm; i += kBlock) {
        fr_store(fo, i, fr_load(tf, i));
        if (TWO) fr_store(go, i, fr_load(tg, i));
    }
    if (W != 0) {
        // reduce every (round, w) vector of parked partials: 8 groups of 32 lanes
        const int grp = tid >> 5, l32 = tid & 31;
        size_t off = 0, m2 = m;
        for (int rd = 0; rd < rounds; rd++) {
            const size_t h = m2 >> 1;
            const size_t cnt = h < (size_t)kBlock ? h : (size_t)kBlock;
            for (int w = 0; w < W; w++) {
                if (((rd * W + w) & 7) == grp) {  // wave-uniform per 32-lane group
                    Fr v = fp_zero<FrCfg>();
                    for (size_t t = l32; t < cnt; t += 32) v = fr_add(v, fr_load(part, off + (size_t)w * cnt + t));
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) {
                        Fr x;
#pragma unroll
                        for (int k = 0; k < 8; k++) x.l[k] = __shfl_down(v.l[k], o, 32);
                        v = fr_add(v, x);
                    }
                    if (l32 == 0) fr_store(sums_out, (size_t)rd * W + w, v);
                }
            }
            off += cnt * W;
            m2 = h;
        }
    }
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
static int ilog2(size_t x) {
    int l = 0;
    while (((size_t)1 << (l + 1)) <= x) l++;
    return l;
}

static size_t pass_blocks(zk_ctx* ctx, size_t m, int k) {
    const size_t q = m >> k;
    size_t blocks = (q + kBlock - 1) / kBlock;
    const size_t maxb = (size_t)ctx->cu_count * 4;
    return blocks > maxb ? maxb : blocks;
}
template <int K, int MODE>
static int launch_pass(zk_ctx* ctx, const void* f, const void* g, void* fo, void* go, size_t m, const uint64_t* chal, void* partials,
                       void* qbase) {
    constexpr int W = ModeTraits<MODE>::W;
    const size_t blocks = pass_blocks(ctx, m, K);
    ChalArgs ch;
    std::memset(&ch, 0, sizeof(ch));
    std::memcpy(&ch, chal, (size_t)K * 32);
    const size_t lds = (W != 0) ? (size_t)K * W * kBlock * 32 : 0;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_pass<K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_pass<K, MODE>), dim3((unsigned)blocks), dim3(kBlock), lds, ctx->stream, f, g, fo, go, m, ch, partials,
                       qbase);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

template <int MODE>
static int run_mode(zk_ctx* ctx, const void* d_f, const void* d_g, size_t len, const uint64_t* h_chal, size_t rounds,
                    uint64_t* h_sums, uint64_t* h_last_f, uint64_t* h_last_g, void* d_out, void* d_q) {
    constexpr int W = ModeTraits<MODE>::W;
    constexpr bool TWO = ModeTraits<MODE>::TWO;
    constexpr int KMAX = 3;
    // tail capacity: tables + parked partial sums must fit the CU's 160 KiB of LDS
    const size_t tail_max = TWO ? 1024 : kTailMax;
    const size_t fr = 32;
    // result block on device: [sums rounds*W][last_f][last_g]
    const size_t res_elems = rounds * W + 2;
    char* d_res = (char*)scratch(ctx, 5, res_elems * fr);
    if (!d_res) return ZK_ERR_OOM;
    void* d_last_f = d_res + rounds * W * fr;
    void* d_last_g = d_res + (rounds * W + 1) * fr;

    const void* cf = d_f;
    const void* cg = d_g;
    size_t m = len, done = 0;
    int flip = 0;
    void* bufs[4] = {nullptr, nullptr, nullptr, nullptr};
    if (len > tail_max && rounds > 0) {
        bufs[0] = scratch(ctx, 0, (len / 2) * fr);
        bufs[1] = scratch(ctx, 1, (len / 4) * fr);
        if (!bufs[0] || !bufs[1]) return ZK_ERR_OOM;
        if (TWO) {
            bufs[2] = scratch(ctx, 2, (len / 2) * fr);
            bufs[3] = scratch(ctx, 3, (len / 4) * fr);
            if (!bufs[2] || !bufs[3]) return ZK_ERR_OOM;
        }
    }
    // plan the passes first: every pass parks its per-block partial sums in its own slice of one arena
    struct Pass {
        int k;
        size_t m, blocks, part_off;
    };
    std::vector<Pass> plan;
    size_t part_bytes = 0;
    {
        size_t mm = m, dd = 0;
        while (dd < rounds && mm > tail_max) {
            const size_t kcap = (MODE == 1) ? 2 : KMAX;  // measured: K = 3 product passes (35 dependent muls per lane) are no faster than K = 2
            const int k = (int)std::min<size_t>({kcap, rounds - dd, (size_t)(ilog2(mm) - ilog2(tail_max))});
            const size_t blocks = pass_blocks(ctx, mm, k);
            plan.push_back(Pass{k, mm, blocks, part_bytes});
            part_bytes += (size_t)k * W * blocks * fr;
            mm >>= k;
            dd += k;
        }
    }
    if ((int)plan.size() > ReducePlan::kMax) return fail(ctx, ZK_ERR_INVALID, "internal: too many passes");
    char* d_part = nullptr;
    if (W != 0 && part_bytes) {
        d_part = (char*)scratch(ctx, 4, part_bytes);
        if (!d_part) return ZK_ERR_OOM;
    }
    ReducePlan rp;
    std::memset(&rp, 0, sizeof(rp));
    for (const Pass& ps : plan) {
        const int k = ps.k;
        const bool final_out = (MODE == 2) && (done + k == rounds);
        void* fo = final_out ? d_out : bufs[flip];
        void* go = TWO ? bufs[2 + flip] : nullptr;
        void* qb = (MODE == 3) ? (char*)d_q + (len - m) * fr : nullptr;
        void* part = d_part ? d_part + ps.part_off : nullptr;
        int rc;
        if (k == 3) rc = launch_pass<3, MODE>(ctx, cf, cg, fo, go, m, h_chal + 4 * done, part, qb);
        else if (k == 2) rc = launch_pass<2, MODE>(ctx, cf, cg, fo, go, m, h_chal + 4 * done, part, qb);
        else rc = launch_pass<1, MODE>(ctx, cf, cg, fo, go, m, h_chal + 4 * done, part, qb);
        if (rc) return rc;
        if (W != 0) {  // outputs of this pass: sums of rounds done .. done+k-1, consecutive in d_res
            rp.partials[rp.n] = part;
            rp.nb[rp.n] = (unsigned)ps.blocks;
            rp.first[rp.n] = (unsigned)(done * W);
            rp.n++;
            rp.first[rp.n] = (unsigned)((done + k) * W);
        }
        cf = fo;
        cg = go;
        m >>= k;
        done += k;
        flip ^= 1;
    }
    if (W != 0 && rp.n) {
        hipLaunchKernelGGL(k_reduce_all, dim3(rp.first[rp.n]), dim3(kBlock), (size_t)kBlock * 32, ctx->stream, rp, (void*)d_res);
        ZK_HIP(ctx, hipGetLastError());
    }
    if (done < rounds || MODE != 2) {
        // tail: the remaining rounds (possibly zero) in one workgroup; also emits the final table
        const int rl = (int)(rounds - done);
        if (m > tail_max) return fail(ctx, ZK_ERR_INVALID, "internal: tail too large");
        size_t parked = 0;  // sum over rounds of W * min(h, 256) partials
        for (size_t mm = m, r2 = 0; r2 < (size_t)rl; r2++, mm >>= 1) parked += (size_t)W * std::min<size_t>(mm >> 1, kBlock);
        size_t lds = (TWO ? 2 : 1) * m * fr + std::max<size_t>(parked, 1) * fr;
        void* fo = (MODE == 2) ? d_out : d_last_f;
        void* qb = (MODE == 3) ? (char*)d_q + (len - m) * fr : nullptr;
        // per call: the attribute belongs to the CURRENT device, and one process may hold a ctx per GPU
        hipFuncSetAttribute((const void*)k_tail<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        TailChal tc;
        std::memset(&tc, 0, sizeof(tc));
        if (rl > 11) return fail(ctx, ZK_ERR_INVALID, "internal: tail rounds");
        if (rl) std::memcpy(tc.c, h_chal + 4 * done, (size_t)rl * fr);
        hipLaunchKernelGGL((k_tail<MODE>), dim3(1), dim3(kBlock), lds, ctx->stream, cf, cg, m, rl, tc,
                           (void*)(d_res + done * W * fr), qb, fo, d_last_g);
        ZK_HIP(ctx, hipGetLastError());
    } else if (rounds == 0) {
        ZK_HIP(ctx, hipMemcpyAsync(d_out, d_f, len * fr, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (MODE != 2) {
        char* h = (char*)pinned(ctx, res_elems * fr);
        if (!h) return ZK_ERR_OOM;
        ZK_HIP(ctx, hipMemcpyAsync(h, d_res, res_elems * fr, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h_sums && rounds * W) std::memcpy(h_sums, h, rounds * W * fr);
        if (h_last_f) std::memcpy(h_last_f, h + rounds * W * fr, fr);
        if (TWO && h_last_g) std::memcpy(h_last_g, h + (rounds * W + 1) * fr, fr);
    }
    return ZK_OK;
}

int multilinear_run(zk_ctx* ctx, int mode, const void* d_f, const void* d_g, size_t len, const uint64_t* h_chal, size_t rounds,
                    uint64_t* h_sums, uint64_t* h_last_f, uint64_t* h_last_g, void* d_out, void* d_q) {
    if (len == 0 || (len & (len - 1))) return fail(ctx, ZK_ERR_INVALID, "table length %zu is not a power of two", len);
    if (rounds > (size_t)ilog2(len)) return fail(ctx, ZK_ERR_INVALID, "more rounds than variables");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    switch (mode) {
        case 0: return run_mode<0>(ctx, d_f, d_g, len, h_chal, rounds, h_sums, h_last_f, h_last_g, d_out, d_q);
        case 1: return run_mode<1>(ctx, d_f, d_g, len, h_chal, rounds, h_sums, h_last_f, h_last_g, d_out, d_q);
        case 2: return run_mode<2>(ctx, d_f, d_g, len, h_chal, rounds, h_sums, h_last_f, h_last_g, d_out, d_q);
        case 3: return run_mode<3>(ctx, d_f, d_g, len, h_chal, rounds, h_sums, h_last_f, h_last_g, d_out, d_q);
    }
    return fail(ctx, ZK_ERR_INVALID, "bad mode");
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
