// zk_msm.hip -- G1 multi-scalar multiplication on gfx950: the `G::msm(bases, scalars)` call of
// d_msm (dist-primitive/src/dmsm.rs:23) and of commit/open (dpoly_comm.rs:242,274,457).
//
// Pipeline (signed-digit Pippenger, window c, W windows, nb = 2^(c-1) buckets per window):
//   1 k_digits   scalars out of Montgomery form (as ark `into_bigint`), signed c-bit digits,
//                per-(window,bucket) histogram                           [coalesced 32-B reads]
//   2 k_scan     exclusive scan of the histogram per window
//   3 k_scatter  counting sort: point indices grouped by bucket          [atomics-free adds later]
//   4 k_accum    one lane per bucket walks its run, gathers 96-B affine bases from HBM and
//                accumulates with XYZZ mixed adds (8M+2S)                [the dominant kernel]
//   5 k_halve    bucket reduction WITHOUT the serial running sum: sum_b b*B_b is split into
//                bit planes.  Each pass pairs neighbours (L[2j]+L[2j+1]) and peels the odd
//                elements off as a new row whose plain sum is the plane T_k; rows keep halving.
//                After c-1 passes every window is down to c points (T_0..T_{c-2}, T_all).
//   6 host       sum_w 2^{cw} (T_all + sum_k 2^k T_k): ~255 doublings, a pure dependency chain
//                (host_curve.hpp), then normalisation to affine.
// All additions are exact group operations, so the affine result is independent of the
// (non-deterministic) order in which the sort places points inside a bucket.
#include "curve.cuh"
#include "host_curve.hpp"
#include "zk_ctx.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

namespace zk {

static constexpr int kBlk = 256;
static constexpr u32 kSkip = 0xffffffffu;

int msm_pick_window(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = lg - 4;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
}

// number of windows so that the top signed digit never carries out (scalars are < r < 2^255)
static int msm_windows(int c) {
    static const u64 Rm1[4] = {0xffffffff00000000ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    int W = (255 + c - 1) / c;
    const int sh = c * (W - 1);  // <= 254
    const int li = sh / 64, bi = sh % 64;
    u64 top = Rm1[li] >> bi;
    if (bi && li + 1 < 4) top |= Rm1[li + 1] << (64 - bi);
    // the top window sees at most top + 1 (carry in) and must stay <= 2^(c-1) to remain positive
    if (top + 1 > ((u64)1 << (c - 1))) W++;
    return W;
}

// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlk) k_digits(const void* __restrict__ scalars, size_t n, int c, int W, size_t nb,
                                               u32* __restrict__ digits, u32* __restrict__ counts) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    Fr s = fp_from_mont<FrCfg>(fr_load(scalars, i));
    const u32 mask = (1u << c) - 1u;
    u32 carry = 0;
    for (int w = 0; w < W; w++) {
        u32 v = (s.l[0] & mask) + carry;
        // s >>= c   (c <= 16 < 32)
#pragma unroll
        for (int k = 0; k < 7; k++) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));
        s.l[7] >>= c;
        u32 d;
        if (v > (u32)nb) {
            v = (1u << c) - v;
            carry = 1;
            d = 0x80000000u;
        } else {
            carry = 0;
            d = 0;
        }
        if (v == 0) {
            d = kSkip;
        } else {
            d |= (v - 1);
            atomicAdd(&counts[(size_t)w * nb + (v - 1)], 1u);
        }
        digits[(size_t)w * n + i] = d;
    }
}

// exclusive scan of counts[w][0..nb) -> offsets, cursor.  One block per window.
__global__ void __launch_bounds__(kBlk) k_scan(const u32* __restrict__ counts, size_t nb, u32* __restrict__ offsets,
                                             u32* __restrict__ cursor) {
    __shared__ u32 part[kBlk];
    const int w = blockIdx.x, tid = threadIdx.x;
    const size_t per = (nb + kBlk - 1) / kBlk;
    const size_t lo = (size_t)tid * per, hi = (lo + per < nb) ? lo + per : nb;
    u32 s = 0;
    for (size_t b = lo; b < hi; b++) s += counts[(size_t)w * nb + b];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (int t = 0; t < kBlk; t++) {
            u32 v = part[t];
            part[t] = run;
            run += v;
        }
    }
    __syncthreads();
    u32 run = part[tid];
    for (size_t b = lo; b < hi; b++) {
        offsets[(size_t)w * nb + b] = run;
        cursor[(size_t)w * nb + b] = run;
        run += counts[(size_t)w * nb + b];
    }
}

__global__ void __launch_bounds__(kBlk) k_scatter(const u32* __restrict__ digits, size_t n, size_t nb, u32* __restrict__ cursor,
                                                u32* __restrict__ sorted) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    const int w = blockIdx.y;
    if (i >= n) return;
    u32 d = digits[(size_t)w * n + i];
    if (d == kSkip) return;
    u32 pos = atomicAdd(&cursor[(size_t)w * nb + (d & 0x7fffffffu)], 1u);
    sorted[(size_t)w * n + pos] = (u32)i | (d & 0x80000000u);
}

// bucket accumulation: lane g = (w, b) sums its run.  buckets: [W][nb] Xyzz
__global__ void __launch_bounds__(kBlk) k_accum(const void* __restrict__ bases, const u32* __restrict__ sorted,
                                              const u32* __restrict__ offsets, const u32* __restrict__ counts, size_t n, size_t nb,
                                              size_t total, void* __restrict__ buckets) {
    const size_t g = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (g >= total) return;
    const size_t w = g / nb;
    const u32* run = sorted + w * n + offsets[g];
    const u32 cnt = counts[g];
    Xyzz acc;
    xyzz_set_inf(acc);
    for (u32 e = 0; e < cnt; e++) {
        u32 v = run[e];
        Aff p = aff_load(bases, v & 0x7fffffffu);
        xyzz_madd(acc, p, (v >> 31) != 0);
    }
    xyzz_store(buckets, g, acc);
}

// one bit-plane pass of the bucket reduction.  in: [W][rows][len], out: [W][rows+1][len/2]
__global__ void __launch_bounds__(kBlk) k_halve(const void* __restrict__ in, void* __restrict__ out, int W, int rows, size_t len) {
    const size_t half = len >> 1;
    const size_t per_w = (size_t)(rows + 1) * half;
    const size_t t = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (t >= per_w * W) return;
    const size_t w = t / per_w, rem = t % per_w;
    const int r = (int)(rem / half);
    const size_t j = rem % half;
    const size_t in_w = w * (size_t)rows * len;
    const size_t src_row = (r < rows - 1) ? (size_t)r : (size_t)(rows - 1);  // rows-1 = the L row
    Xyzz b = xyzz_load(in, in_w + src_row * len + 2 * j + 1);
    Xyzz res;
    if (r == rows - 1) {
        res = b;  // odd elements of L become the new plane row
    } else {
        Xyzz a = xyzz_load(in, in_w + src_row * len + 2 * j);
        res = xyzz_add(a, b);
    }
    xyzz_store(out, t, res);
}

// test hook: XYZZ arithmetic on pairs of affine points
__global__ void __launch_bounds__(kBlk) k_dbg_g1(const void* __restrict__ p, const void* __restrict__ q, void* __restrict__ out,
                                               size_t n, int mode) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    Aff a = aff_load(p, i), b = aff_load(q, i);
    Xyzz s;
    xyzz_set_inf(s);
    xyzz_madd(s, a, false);
    xyzz_madd(s, b, mode == 3);  // p + q   (mode 3: p - q)
    Xyzz r = s;
    if (mode == 1) {  // (p+q) + p
        Xyzz pa;
        xyzz_set_inf(pa);
        xyzz_madd(pa, a, false);
        r = xyzz_add(s, pa);
    } else if (mode == 2) {  // (p+q) + (p+q): doubling path of the full addition
        r = xyzz_add(s, s);
    }
    xyzz_store(out, i, r);
}

// ---------------------------------------------------------------------------------------
static zkhost::Jac load_xyzz_host(const uint64_t* p) {
    zkhost::Fq X, Y, ZZ, ZZZ;
    std::memcpy(X.data(), p, 48);
    std::memcpy(Y.data(), p + 6, 48);
    std::memcpy(ZZ.data(), p + 12, 48);
    std::memcpy(ZZZ.data(), p + 18, 48);
    return zkhost::xyzz_to_jac(X, Y, ZZ, ZZZ);
}

int msm_g1(zk_ctx* ctx, const zk_srs* srs, size_t offset, const void* d_scalars, size_t n, uint64_t* h_out) {
    if (!srs || !h_out) return fail(ctx, ZK_ERR_INVALID, "null argument");
    if (offset + n > srs->n) return fail(ctx, ZK_ERR_LENGTH, "msm: %zu scalars but only %zu bases from offset %zu", n, srs->n - std::min(offset, srs->n), offset);
    if (n >= ((size_t)1 << 31)) return fail(ctx, ZK_ERR_INVALID, "msm: n too large");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) {
        zkhost::write_normalised(zkhost::jac_inf(), h_out);
        return ZK_OK;
    }
    const int c = ctx->msm_window_override > 0 ? ctx->msm_window_override : msm_pick_window(n);
    const int W = msm_windows(c);
    const size_t nb = (size_t)1 << (c - 1);
    const size_t total = (size_t)W * nb;
    hipStream_t st = ctx->stream;

    u32* digits = (u32*)scratch(ctx, 0, (size_t)W * n * 4);
    u32* sorted = (u32*)scratch(ctx, 1, (size_t)W * n * 4);
    u32* cnts = (u32*)scratch(ctx, 2, 3 * total * 4);
    void* bufA = scratch(ctx, 3, total * 192);
    void* bufB = scratch(ctx, 4, total * 192);
    if (!digits || !sorted || !cnts || !bufA || !bufB) return ZK_ERR_OOM;
    u32* counts = cnts;
    u32* offsets = cnts + total;
    u32* cursor = cnts + 2 * total;
    const char* bases = (const char*)srs->d_bases + offset * 96;

    hipEventRecord(ctx->ev[0], st);
    ZK_HIP(ctx, hipMemsetAsync(counts, 0, total * 4, st));
    const unsigned gn = (unsigned)((n + kBlk - 1) / kBlk);
    hipLaunchKernelGGL(k_digits, dim3(gn), dim3(kBlk), 0, st, d_scalars, n, c, W, nb, digits, counts);
    hipLaunchKernelGGL(k_scan, dim3(W), dim3(kBlk), 0, st, counts, nb, offsets, cursor);
    hipLaunchKernelGGL(k_scatter, dim3(gn, W), dim3(kBlk), 0, st, digits, n, nb, cursor, sorted);
    hipEventRecord(ctx->ev[1], st);
    hipLaunchKernelGGL(k_accum, dim3((unsigned)((total + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, (const void*)bases, sorted, offsets,
                       counts, n, nb, total, bufA);
    hipEventRecord(ctx->ev[2], st);
    void* in = bufA;
    void* out = bufB;
    int rows = 1;
    size_t len = nb;
    while (len > 1) {
        size_t threads = (size_t)W * (rows + 1) * (len >> 1);
        hipLaunchKernelGGL(k_halve, dim3((unsigned)((threads + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, (const void*)in, out, W, rows, len);
        std::swap(in, out);
        rows++;
        len >>= 1;
    }
    ZK_HIP(ctx, hipGetLastError());
    // W * rows points (rows == c), 192 B each
    const size_t npts = (size_t)W * rows;
    uint64_t* h = (uint64_t*)pinned(ctx, npts * 192);
    if (!h) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipMemcpyAsync(h, in, npts * 192, hipMemcpyDeviceToHost, st));
    hipEventRecord(ctx->ev[3], st);
    ZK_HIP(ctx, hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    // host combine: position p = c*w + k carries weight 2^p
    std::vector<zkhost::Jac> pos((size_t)W * c + 1, zkhost::jac_inf());
    for (int w = 0; w < W; w++) {
        for (int k = 0; k < rows; k++) {
            zkhost::Jac pt = load_xyzz_host(h + ((size_t)w * rows + k) * 24);
            if (zkhost::is_zero(pt.z)) continue;
            // rows 0..rows-2 are planes T_k (weight 2^k); the last row is T_all (weight 1)
            size_t p = (size_t)c * w + ((k == rows - 1) ? 0 : k);
            pos[p] = zkhost::jac_add(pos[p], pt);
        }
    }
    zkhost::Jac acc = zkhost::jac_inf();
    for (size_t p = pos.size(); p-- > 0;) {
        acc = zkhost::jac_dbl(acc);
        acc = zkhost::jac_add(acc, pos[p]);
    }
    zkhost::write_normalised(acc, h_out);
    auto t1 = std::chrono::steady_clock::now();
    float ms;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
    ctx->msm_ms[0] = ms;
    hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]);
    ctx->msm_ms[1] = ms;
    hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]);
    ctx->msm_ms[2] = ms;
    ctx->msm_ms[3] = std::chrono::duration<float, std::milli>(t1 - t0).count();
    ctx->msm_ms[4] = ctx->msm_ms[0] + ctx->msm_ms[1] + ctx->msm_ms[2] + ctx->msm_ms[3];
    return ZK_OK;
}

// ---------------------------------------------------------------------------------------
int srs_pack(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out) {
    if (!out || (n && !h_bases)) return fail(ctx, ZK_ERR_INVALID, "null argument");
    if (stride != 96 && stride < 97) return fail(ctx, ZK_ERR_INVALID, "stride must be 96 or >= 97 (x, y, infinity flag)");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    zk_srs* s = new zk_srs();
    s->n = n;
    s->owned = true;
    if (n) {
        ZK_HIP(ctx, hipMalloc(&s->d_bases, n * 96));
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
    }
    *out = s;
    return ZK_OK;
}

int srs_generate(zk_ctx* ctx, const uint64_t* k0, const uint64_t* k1, size_t n, zk_srs** out) {
    namespace H = zkhost;
    if (!out) return fail(ctx, ZK_ERR_INVALID, "null argument");
    H::Aff G{H::to_mont(H::GX_CANON), H::to_mont(H::GY_CANON)};
    H::Aff start = H::jac_to_aff(H::scalar_mul(G, k0));
    H::Aff step = H::jac_to_aff(H::scalar_mul(G, k1));
    std::vector<H::Aff> pts(n);
    const size_t CH = 4096;
    std::vector<H::Jac> chunk;
    H::Jac cur = H::aff_inf(start) ? H::jac_inf() : H::Jac{start.x, start.y, H::ONE};
    for (size_t base = 0; base < n; base += CH) {
        size_t m = std::min(CH, n - base);
        chunk.resize(m);
        for (size_t i = 0; i < m; i++) {
            chunk[i] = cur;
            cur = H::jac_add_mixed(cur, step);
        }
        H::batch_to_affine(chunk, &pts[base]);
    }
    return srs_pack(ctx, pts.data(), 96, n, out);
}

// K9: tiny public linear maps on points (PSS unpack2 / pack of d_msm's leader closure,
// dmsm.rs:30-39; sums of N_p commitments dpoly_comm.rs:289-292): sum_i k_i * P_i for a handful
// of points.  A joint double-and-add on the host: it is a ~255-step dependency chain, the same
// shape as the MSM's final combine.
int g1_lincomb_host(zk_ctx* ctx, const uint64_t* h_points_jac, const uint64_t* h_scalars_canon, size_t n, uint64_t* h_out) {
    namespace H = zkhost;
    std::vector<H::Aff> pts(n);
    std::vector<H::Jac> tmp(n);
    for (size_t i = 0; i < n; i++) {
        std::memcpy(tmp[i].x.data(), h_points_jac + 18 * i, 48);
        std::memcpy(tmp[i].y.data(), h_points_jac + 18 * i + 6, 48);
        std::memcpy(tmp[i].z.data(), h_points_jac + 18 * i + 12, 48);
    }
    if (n) H::batch_to_affine(tmp, pts.data());
    H::Jac acc = H::jac_inf();
    for (int b = 255; b >= 0; b--) {
        acc = H::jac_dbl(acc);
        for (size_t i = 0; i < n; i++)
            if ((h_scalars_canon[4 * i + b / 64] >> (b % 64)) & 1) acc = H::jac_add_mixed(acc, pts[i]);
    }
    H::write_normalised(acc, h_out);
    (void)ctx;
    return ZK_OK;
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

}  // namespace zk
