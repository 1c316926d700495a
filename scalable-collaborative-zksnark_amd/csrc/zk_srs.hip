// zk_srs.hip -- base vectors built ON the device.
//
//   srs_generate          the synthetic SRS of the benchmarks, P_i = (k0 + i k1) G -- the stand-in for the
//                         reference's random-point parameters (PolynomialCommitmentCub::new_single /
//                         new_random, dist-primitive/src/dpoly_comm.rs:197-233).
//   xyzz_to_affine_batch  XYZZ array -> the packed 96-byte affine records every MSM reads (the reference's
//                         `mature()`, dpoly_comm.rs:141-151: projective -> affine), Montgomery's trick per lane
//                         over strided points, prefix products parked in HBM.
//
// An arithmetic sequence of n points is filled level by level: T lanes each own the points {t + i T} and walk
// them with one mixed addition of the constant affine point T*(k1 G) per step (coalesced stores: lane t writes
// index t + i T); their T starting points are the same sequence at a 64x smaller size.  The first <= 1024
// points come from the host.  2^24 points: ~10 + 16 Fq multiplications each (walk + normalisation).
#include "curve30.cuh"
#include "host_curve.hpp"
#include "zk_ctx.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace zk {

static constexpr int kBlk = 256;

struct Aff30Arg {  // an affine point in the internal form, as a kernel argument
    u32 x[13], y[13];
};

__global__ void __launch_bounds__(kBlk) k_aff_to_xyzz(const void* __restrict__ aff96, size_t n, void* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    Aff30 p = aff30_load(aff96, i);
    Xyzz30 r;
    xyzz30_set_inf(r);
    xyzz30_madd(r, p, false);
    xyzz30_store(out, i, r);
}

// lane t: out[t + i T] = starts[t] + i * step, i >= 0, for the indices below n
__global__ void __launch_bounds__(kBlk) k_seq_walk(const void* __restrict__ starts, size_t T, Aff30Arg stepT, size_t n, void* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (t >= T || t >= n) return;
    Aff30 step;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        step.x.l[k] = stepT.x[k];
        step.y.l[k] = stepT.y[k];
    }
    Xyzz30 acc = xyzz30_load(starts, t);
    for (size_t idx = t; idx < n; idx += T) {
        xyzz30_store(out, idx, acc);
        if (idx + T < n) xyzz30_madd(acc, step, false);
    }
}

// XYZZ -> affine (internal Montgomery form, canonical; infinity = (0, 0)).  Lane t owns the points {t + i T}.
// x = X / ZZ = X (ZZ / ZZZ)^2, y = Y / ZZZ: only the ZZZ are inverted.
__global__ void __launch_bounds__(kBlk) k_batch_affine(const void* __restrict__ in, size_t n, size_t T, void* __restrict__ prefix,
                                                     void* __restrict__ out96) {
    const size_t t = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (t >= T || t >= n) return;
    const size_t cnt = (n - t + T - 1) / T;
    Fq30 p = f30_one();
    for (size_t i = 0; i < cnt; i++) {
        const size_t idx = t + i * T;
        f30_store(prefix, idx * 48, p);  // product of the ZZZ before this point (< 2q, fits 48 bytes)
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
        f30_store(out96, idx * 96, x);
        f30_store(out96, idx * 96 + 48, y);
    }
}

// ---- host side ----
static const zkhost::Fq& k64_mont() {  // 64 in the 2^384 Montgomery form: mul(x R, 64 R) = 64 x R = x 2^390 (mod q)
    static const zkhost::Fq k = zkhost::to_mont(zkhost::Fq{64, 0, 0, 0, 0, 0});
    return k;
}
void host_aff_to_internal(const zkhost::Aff& p, uint64_t out[12]) {
    const zkhost::Fq x = zkhost::mul(p.x, k64_mont()), y = zkhost::mul(p.y, k64_mont());
    std::memcpy(out, x.data(), 48);
    std::memcpy(out + 6, y.data(), 48);
}
static Aff30Arg to_arg(const zkhost::Aff& p) {
    uint64_t w[12];
    host_aff_to_internal(p, w);
    Aff30Arg a;
    auto limb = [](const uint64_t* v, int i) -> u32 {  // 30-bit limb i of a 384-bit little-endian integer
        const int bit = 30 * i, wi = bit >> 6, s = bit & 63;
        uint64_t x = v[wi] >> s;
        if (s > 34 && wi + 1 < 6) x |= v[wi + 1] << (64 - s);
        return (u32)(i < 12 ? (x & 0x3fffffffu) : x);
    };
    for (int i = 0; i < 13; i++) {
        a.x[i] = limb(w, i);
        a.y[i] = limb(w + 6, i);
    }
    return a;
}

// normalise n XYZZ points (blocked layout) into packed affine records; d_out96 may not alias d_xyzz
int xyzz_to_affine_batch(zk_ctx* ctx, const void* d_xyzz, size_t n, void* d_out96) {
    if (n == 0) return ZK_OK;
    void* prefix = nullptr;
    ZK_HIP(ctx, device_alloc(ctx, &prefix, n * 48));
    const size_t T = std::max<size_t>(1, (n + 63) / 64);
    hipLaunchKernelGGL(k_batch_affine, dim3((unsigned)((T + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, d_xyzz, n, T, prefix, d_out96);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(prefix);
    if (e != hipSuccess) return hip_fail(ctx, e, "k_batch_affine");
    return ZK_OK;
}

// d_out96[i] = start + i * step (affine, internal form), i < n
int fill_sequence_affine(zk_ctx* ctx, const zkhost::Aff& start, const zkhost::Aff& step, size_t n, void* d_out96) {
    namespace H = zkhost;
    if (n == 0) return ZK_OK;
    std::vector<size_t> sizes{n};
    while (sizes.back() > 1024) sizes.push_back(std::min(sizes.back(), std::max<size_t>(1024, (sizes.back() + 63) / 64)));
    // the smallest level on the host
    const size_t m0 = sizes.back();
    std::vector<uint64_t> h0(m0 * 12);
    {
        std::vector<H::Jac> seq(m0);
        H::Jac cur = H::aff_inf(start) ? H::jac_inf() : H::Jac{start.x, start.y, H::ONE};
        for (size_t i = 0; i < m0; i++) {
            seq[i] = cur;
            cur = H::jac_add_mixed(cur, step);
        }
        std::vector<H::Aff> aff(m0);
        H::batch_to_affine(seq, aff.data());
        for (size_t i = 0; i < m0; i++) host_aff_to_internal(aff[i], &h0[12 * i]);
    }
    auto blocks64 = [](size_t k) { return ((k + 63) & ~(size_t)63) * 192; };
    void *bufA = nullptr, *bufB = nullptr, *d_h0 = nullptr;
    auto cleanup = [&] {
        if (bufA) hipFree(bufA);
        if (bufB) hipFree(bufB);
        if (d_h0) hipFree(d_h0);
    };
    hipError_t e = device_alloc(ctx, &bufA, blocks64(n));
    if (e == hipSuccess) e = device_alloc(ctx, &bufB, blocks64(sizes.size() > 1 ? sizes[1] : m0));
    if (e == hipSuccess) e = device_alloc(ctx, &d_h0, m0 * 96);
    if (e == hipSuccess) e = hipMemcpyAsync(d_h0, h0.data(), m0 * 96, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        return hip_fail(ctx, e, "fill_sequence_affine: allocation");
    }
    // levels are numbered from the top (0 = n points); level k is written to bufA when k is even
    const int L = (int)sizes.size() - 1;
    void* cur = (L % 2 == 0) ? bufA : bufB;
    hipLaunchKernelGGL(k_aff_to_xyzz, dim3((unsigned)((m0 + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)d_h0, m0, cur);
    for (int k = L - 1; k >= 0; k--) {
        const size_t T = sizes[k + 1], nk = sizes[k];
        uint64_t ts[4] = {T, 0, 0, 0};
        const Aff30Arg stepT = to_arg(H::jac_to_aff(H::scalar_mul(step, ts)));
        void* nxt = (k % 2 == 0) ? bufA : bufB;
        hipLaunchKernelGGL(k_seq_walk, dim3((unsigned)((T + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)cur, T, stepT, nk, nxt);
        cur = nxt;
    }
    e = hipGetLastError();
    int rc = (e == hipSuccess) ? xyzz_to_affine_batch(ctx, cur, n, d_out96) : hip_fail(ctx, e, "k_seq_walk");
    hipStreamSynchronize(ctx->stream);
    cleanup();
    return rc;
}

// ---------------------------------------------------------------------------------------
// K9 on points: a small PUBLIC Fr matrix applied to k vectors of G1 points at once -- the packed-secret-sharing
// maps on group elements (pack_from_public / unpack / unpack2 are generic over DomainCoeff, secret-sharing/src/
// pss.rs:93-171; the reference runs them as group FFTs on the leader, 48 ms per item in its sample log, and
// over every l-chunk of the SRS in PolynomialCommitmentCub::to_packed, dpoly_comm.rs:164-194):
//   out[j*osv + r*osr] = sum_c M[r][c] * P[j*isv + c*isc]
// One lane per output point: a joint double-and-add over the `cols` points of its vector (the scalars are
// public and shared by all vectors, so every lane of a row follows the same bit pattern: no divergence).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlk) k_g1_apply_matrix(const u32* __restrict__ M, size_t rows, size_t cols, int topbit,
                                                        const void* __restrict__ in, size_t isv, size_t isc, size_t k,
                                                        void* __restrict__ out_xyzz) {
    const size_t t = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (t >= rows * k) return;
    const size_t r = t / k, j = t % k;  // consecutive lanes -> consecutive vectors of the same row
    const u32* Mr = M + r * cols * 8;
    Xyzz30 acc;
    xyzz30_set_inf(acc);
    for (int b = topbit; b >= 0; b--) {
        acc = xyzz30_dbl(acc);
        for (size_t c = 0; c < cols; c++)
            if ((Mr[c * 8 + (b >> 5)] >> (b & 31)) & 1u) xyzz30_madd(acc, aff30_load(in, j * isv + c * isc), false);
    }
    xyzz30_store(out_xyzz, t, acc);
}
// The same map with ONE LANE PER (output, column): every lane is a single scalar multiplication M[r][c] * P_c, the `cols` terms
// of an output are then added pairwise (k_xyzz_pair_sums, log2(cols) tiny launches).  ~3x the point operations of the joint form
// (no shared doublings), but cols times the parallelism: the large-l maps of the leader -- unpack2 on 8l shares into l secrets
// for a handful of vectors (dmsm.rs:30-39 at l = 16: 16 outputs of 128 terms each) -- are latency chains of 255 doublings + ~128
// additions instead of 255 + ~16 000.  terms[(t * cols) + c], t = r * k + j.
__global__ void __launch_bounds__(kBlk) k_g1_scale_cols(const u32* __restrict__ M, size_t rows, size_t cols, int topbit, const void* __restrict__ in,
                                                      size_t isv, size_t isc, size_t k, void* __restrict__ terms) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= rows * k * cols) return;
    const size_t c = i % cols, t = i / cols, r = t / k, j = t % k;
    const u32* m = M + (r * cols + c) * 8;
    const Aff30 p = aff30_load(in, j * isv + c * isc);
    Xyzz30 acc;
    xyzz30_set_inf(acc);
    for (int b = topbit; b >= 0; b--) {
        acc = xyzz30_dbl(acc);
        if ((m[b >> 5] >> (b & 31)) & 1u) xyzz30_madd(acc, p, false);
    }
    xyzz30_store(terms, i, acc);
}
// out[t * half + h] = in[t * width + 2h] + in[t * width + 2h + 1] (the odd one out is copied), half = ceil(width / 2)
__global__ void __launch_bounds__(kBlk) k_xyzz_pair_sums(const void* __restrict__ in, size_t n, size_t width, void* __restrict__ out) {
    const size_t half = (width + 1) / 2, i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n * half) return;
    const size_t t = i / half, h = i % half;
    const Xyzz30 a = xyzz30_load(in, t * width + 2 * h);
    xyzz30_store(out, i, (2 * h + 1 < width) ? xyzz30_add(a, xyzz30_load(in, t * width + 2 * h + 1)) : a);
}
__global__ void __launch_bounds__(kBlk) k_copy96_strided(const void* __restrict__ in, void* __restrict__ out, size_t rows, size_t k,
                                                       size_t osv, size_t osr) {
    const size_t t = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (t >= rows * k) return;
    const size_t r = t / k, j = t % k;
    const uint4* s = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(in) + t * 96);
    uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (j * osv + r * osr) * 96);
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = s[i];
}

// points and results in the INTERNAL affine form (SRS device copies); h_matrix: rows*cols canonical 4 x u64 scalars
int g1_apply_matrix_internal(zk_ctx* ctx, const uint64_t* h_matrix, size_t rows, size_t cols, const void* d_in, size_t isv, size_t isc,
                             void* d_out, size_t osv, size_t osr, size_t k) {
    if (rows == 0 || k == 0) return ZK_OK;
    int top = -1;
    for (size_t i = 0; i < rows * cols; i++)
        for (int b = 255; b > top; b--)
            if ((h_matrix[4 * i + b / 64] >> (b % 64)) & 1) {
                top = b;
                break;
            }
    const size_t total = rows * k;
    // few outputs of many terms each: one lane per term (see k_g1_scale_cols); otherwise one lane per output
    // (capped at 2^18 terms = 48 + 24 MiB of transient scratch: the motivating case is the leader's maps of 3 x 16 outputs of 128 terms; a GPU
    // shared by several parties must not be asked for hundreds of MiB per call.  If even that is refused, the row form -- total * 192 B -- runs.)
    bool by_column = cols >= 8 && total < 16384 && total * cols <= ((size_t)1 << 18) && tuning().g1_map_by_column != 0;
    void *d_m = nullptr, *d_x = nullptr, *d_a = nullptr, *d_t = nullptr;
    auto cleanup = [&] {
        if (d_m) hipFree(d_m);
        if (d_x) hipFree(d_x);
        if (d_a) hipFree(d_a);
        if (d_t) hipFree(d_t);
    };
    hipError_t e = device_alloc(ctx, &d_m, std::max<size_t>(rows * cols * 32, 32));
    if (e == hipSuccess && by_column) {
        hipError_t ec = device_alloc(ctx, &d_x, ((total * ((cols + 1) / 2) + 63) & ~(size_t)63) * 192);
        if (ec == hipSuccess) ec = device_alloc(ctx, &d_t, ((total * cols + 63) & ~(size_t)63) * 192);
        if (ec != hipSuccess) {  // out of memory for the per-term scratch: fall back to one lane per output
            (void)hipGetLastError();
            if (d_x) hipFree(d_x);
            if (d_t) hipFree(d_t);
            d_x = d_t = nullptr;
            by_column = false;
        }
    }
    if (e == hipSuccess && !by_column) e = device_alloc(ctx, &d_x, ((total + 63) & ~(size_t)63) * 192);
    if (e == hipSuccess) e = device_alloc(ctx, &d_a, total * 96);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m, h_matrix, rows * cols * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // h_matrix is caller memory
    if (e != hipSuccess) {
        cleanup();
        return hip_fail(ctx, e, "g1_apply_matrix: allocation");
    }
    void* d_sum = d_x;
    if (by_column) {
        hipLaunchKernelGGL(k_g1_scale_cols, dim3((unsigned)((total * cols + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const u32*)d_m, rows, cols, top,
                           d_in, isv, isc, k, d_t);
        void *src = d_t, *dst = d_x;  // (both hold ceil(width / 2) terms per output from the first pass on)
        for (size_t width = cols; width > 1; width = (width + 1) / 2) {
            hipLaunchKernelGGL(k_xyzz_pair_sums, dim3((unsigned)((total * ((width + 1) / 2) + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)src, total, width,
                               dst);
            std::swap(src, dst);
        }
        d_sum = src;
    } else {
        hipLaunchKernelGGL(k_g1_apply_matrix, dim3((unsigned)((total + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const u32*)d_m, rows, cols, top,
                           d_in, isv, isc, k, d_x);
    }
    int rc = xyzz_to_affine_batch(ctx, d_sum, total, d_a);
    if (!rc) {
        hipLaunchKernelGGL(k_copy96_strided, dim3((unsigned)((total + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)d_a, d_out, rows, k,
                           osv, osr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "k_copy96_strided");
    }
    cleanup();
    return rc;
}

void srs_convert_form(zk_ctx* ctx, const void* in, void* out, size_t npoints, bool to_internal);

// the same on device buffers in the REFERENCE layout (96-byte affine records, Montgomery radix 2^384)
int g1_apply_matrix_ref(zk_ctx* ctx, const uint64_t* h_matrix, size_t rows, size_t cols, const void* d_in, size_t isv, size_t isc, void* d_out,
                        size_t osv, size_t osr, size_t k) {
    if (rows == 0 || k == 0) return ZK_OK;
    const size_t span = (k - 1) * isv + (cols ? (cols - 1) * isc : 0) + 1, total = rows * k;
    void *d_i = nullptr, *d_o = nullptr;
    hipError_t e = device_alloc(ctx, &d_i, span * 96);
    if (e == hipSuccess) e = device_alloc(ctx, &d_o, total * 96);
    int rc = ZK_OK;
    if (e != hipSuccess) rc = hip_fail(ctx, e, "zk_g1_apply_matrix: allocation");
    if (!rc) {
        srs_convert_form(ctx, d_in, d_i, span, true);
        rc = g1_apply_matrix_internal(ctx, h_matrix, rows, cols, d_i, isv, isc, d_o, 1, k, k);
    }
    if (!rc) {
        srs_convert_form(ctx, d_o, d_o, total, false);
        hipLaunchKernelGGL(k_copy96_strided, dim3((unsigned)((total + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)d_o, d_out, rows, k, osv, osr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "zk_g1_apply_matrix");
    }
    if (d_i) hipFree(d_i);
    if (d_o) hipFree(d_o);
    return rc;
}

// ---------------------------------------------------------------------------------------
// Structured SRS, PolynomialCommitmentCub::new (dpoly_comm.rs:37-67): powers_of_g[0] = [g] and
//   powers_of_g[i+1] = [e * (1 - s_{n-i-1}) for e in powers_of_g[i]] ++ [e * s_{n-i-1} for e in powers_of_g[i]],
// i.e. level k holds g^{E_k[j]} with E_0 = [1], E_{k+1} = E_k (1 - s) ++ E_k s.  The reference scalar-multiplies
// every point of a level twice (2^(k+1) full 255-bit multiplications); here the exponents are expanded in Fr on the
// device and each point is ONE fixed-base multiplication: 32 mixed additions against a table of d * 2^{8j} * g.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlk) k_eq_expand(const void* __restrict__ in, size_t h, Fr s, Fr one_minus_s, void* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (j >= h) return;
    const Fr e = fr_load(in, j);
    fr_store(out, j, fr_mul(e, one_minus_s));
    fr_store(out, j + h, fr_mul(e, s));
}
// table[j * 255 + (d - 1)] = d * 2^{8j} * g (affine, internal form), j < 32, 1 <= d <= 255
__global__ void __launch_bounds__(kBlk) k_fixed_base_mul(const void* __restrict__ table, const void* __restrict__ scalars, size_t n,
                                                       void* __restrict__ out_xyzz) {
    const size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    const Fr e = fp_from_mont<FrCfg>(fr_load(scalars, i));
    Xyzz30 acc;
    xyzz30_set_inf(acc);
    for (int j = 0; j < 32; j++) {
        const u32 d = (e.l[j >> 2] >> (8 * (j & 3))) & 0xffu;
        if (d) xyzz30_madd(acc, aff30_load(table, (size_t)j * 255 + (d - 1)), false);
    }
    xyzz30_store(out_xyzz, i, acc);
}

static void host_fr_to_arg(const uint64_t* limbs, Fr& out) { std::memcpy(out.l, limbs, 32); }

// h_table: 32 * 255 affine points of the fixed base in the internal form
static void build_fixed_base_table(const zkhost::Aff& g, std::vector<uint64_t>& h_table) {
    namespace H = zkhost;
    std::vector<H::Jac> pts(32 * 255);
    H::Jac basej = H::aff_inf(g) ? H::jac_inf() : H::Jac{g.x, g.y, H::ONE};
    for (int j = 0; j < 32; j++) {
        const H::Aff b = H::jac_to_aff(basej);
        H::Jac cur = basej;
        for (int d = 1; d <= 255; d++) {
            pts[(size_t)j * 255 + d - 1] = cur;
            cur = H::jac_add_mixed(cur, b);
        }
        for (int k = 0; k < 8; k++) basej = H::jac_dbl(basej);
    }
    std::vector<H::Aff> aff(pts.size());
    H::batch_to_affine(pts, aff.data());
    h_table.resize(pts.size() * 12);
    for (size_t i = 0; i < aff.size(); i++) host_aff_to_internal(aff[i], &h_table[12 * i]);
}

// d_out96[i] = scalars[i] * g (internal affine form); d_scalars: n Fr in Montgomery form on the device
int fixed_base_mul(zk_ctx* ctx, const void* d_table, const void* d_scalars, size_t n, void* d_out96) {
    if (n == 0) return ZK_OK;
    void* d_x = nullptr;
    ZK_HIP(ctx, device_alloc(ctx, &d_x, ((n + 63) & ~(size_t)63) * 192));
    hipLaunchKernelGGL(k_fixed_base_mul, dim3((unsigned)((n + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, d_table, d_scalars, n, d_x);
    int rc = xyzz_to_affine_batch(ctx, d_x, n, d_out96);
    hipFree(d_x);
    return rc;
}

// finishes a zk_srs whose first n affine records (internal form) are in place: the endomorphism images
void srs_finish_endo(zk_ctx* ctx, zk_srs* s);

int srs_powers(zk_ctx* ctx, const void* h_g96, const uint64_t* h_s, size_t nvars, zk_srs** out_levels) {
    namespace H = zkhost;
    if (!out_levels || (nvars && !h_s)) return fail(ctx, ZK_ERR_INVALID, "null argument");
    if (nvars > 30) return fail(ctx, ZK_ERR_INVALID, "zk_srs_powers: more than 30 variables");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    H::Aff g{H::to_mont(H::GX_CANON), H::to_mont(H::GY_CANON)};
    if (h_g96) {
        std::memcpy(g.x.data(), h_g96, 48);
        std::memcpy(g.y.data(), (const char*)h_g96 + 48, 48);
    }
    std::vector<uint64_t> h_table;
    build_fixed_base_table(g, h_table);
    const size_t nmax = (size_t)1 << nvars;
    void *d_table = nullptr, *d_e[2] = {nullptr, nullptr};
    std::vector<zk_srs*> made;
    auto cleanup = [&](bool drop_levels) {
        if (d_table) hipFree(d_table);
        if (d_e[0]) hipFree(d_e[0]);
        if (d_e[1]) hipFree(d_e[1]);
        if (drop_levels)
            for (zk_srs* s : made) {
                if (s->d_bases) hipFree(s->d_bases);
                delete s;
            }
    };
    hipError_t e = device_alloc(ctx, &d_table, h_table.size() * 8);
    if (e == hipSuccess) e = device_alloc(ctx, &d_e[0], nmax * 32);
    if (e == hipSuccess) e = device_alloc(ctx, &d_e[1], nmax * 32);
    if (e == hipSuccess) e = hipMemcpyAsync(d_table, h_table.data(), h_table.size() * 8, hipMemcpyHostToDevice, ctx->stream);
    const uint64_t one_m[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};  // R mod r
    if (e == hipSuccess) e = hipMemcpyAsync(d_e[0], one_m, 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        cleanup(true);
        return hip_fail(ctx, e, "zk_srs_powers: allocation");
    }
    int cur = 0;
    for (size_t k = 0; k <= nvars; k++) {
        const size_t len = (size_t)1 << k;
        if (k > 0) {  // E_k = E_{k-1} (1 - s) ++ E_{k-1} s with s = s_{n-k}
            const uint64_t* sk = h_s + 4 * (nvars - k);
            // 1 - s in Montgomery form on the host: R - s (mod r)
            static const uint64_t RM[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
            uint64_t d[4];
            {
                H::u128 bw = 0;
                for (int i = 0; i < 4; i++) {
                    H::u128 t = (H::u128)one_m[i] - sk[i] - (uint64_t)bw;
                    d[i] = (uint64_t)t;
                    bw = (t >> 64) & 1;
                }
                if (bw) {
                    H::u128 c = 0;
                    for (int i = 0; i < 4; i++) {
                        c += (H::u128)d[i] + RM[i];
                        d[i] = (uint64_t)c;
                        c >>= 64;
                    }
                }
            }
            Fr fs, f1s;
            host_fr_to_arg(sk, fs);
            host_fr_to_arg(d, f1s);
            const size_t h = len >> 1;
            hipLaunchKernelGGL(k_eq_expand, dim3((unsigned)((h + kBlk - 1) / kBlk)), dim3(kBlk), 0, ctx->stream, (const void*)d_e[cur], h, fs, f1s,
                               d_e[cur ^ 1]);
            cur ^= 1;
        }
        zk_srs* s = new zk_srs();
        s->n = len;
        s->owned = true;
        made.push_back(s);
        e = device_alloc(ctx, &s->d_bases, 2 * len * 96);
        int rc = (e == hipSuccess) ? fixed_base_mul(ctx, d_table, d_e[cur], len, s->d_bases) : hip_fail(ctx, e, "zk_srs_powers: level allocation");
        if (rc) {
            cleanup(true);
            return rc;
        }
        srs_finish_endo(ctx, s);
    }
    e = hipStreamSynchronize(ctx->stream);
    cleanup(e != hipSuccess);
    if (e != hipSuccess) return hip_fail(ctx, e, "zk_srs_powers");
    for (size_t k = 0; k <= nvars; k++) out_levels[k] = made[k];
    return ZK_OK;
}

// one party's packed level (to_packed, dpoly_comm.rs:164-194): out[k] = sum_{j<l} row[j] * P[k l + j]; a level
// shorter than l is zero-extended to one chunk (:177-181)
int srs_to_packed(zk_ctx* ctx, const zk_srs* level, const uint64_t* h_row, size_t l, zk_srs** out) {
    if (!level || !h_row || !out || l == 0) return fail(ctx, ZK_ERR_INVALID, "null argument");
    if (level->g2) return fail(ctx, ZK_ERR_INVALID, "zk_srs_to_packed: G1 levels only");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = level->n;
    const size_t cols = std::min(l, n), k = n < l ? (n ? 1 : 0) : n / l;
    zk_srs* s = new zk_srs();
    s->n = k;
    s->owned = true;
    if (k) {
        hipError_t e = device_alloc(ctx, &s->d_bases, 2 * k * 96);
        int rc = (e == hipSuccess) ? g1_apply_matrix_internal(ctx, h_row, 1, cols, level->d_bases, cols, 1, s->d_bases, 1, k, k) : hip_fail(ctx, e, "zk_srs_to_packed");
        if (rc) {
            if (s->d_bases) hipFree(s->d_bases);
            delete s;
            return rc;
        }
        srs_finish_endo(ctx, s);
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = s;
    return ZK_OK;
}

}  // namespace zk
