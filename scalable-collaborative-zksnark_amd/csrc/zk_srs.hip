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

}  // namespace zk
