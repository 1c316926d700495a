// RAII wrappers over the C ABI of libzkhip.so (include/zkhip.h): the compute backend of the C++ host mirror.
// Nothing here computes: every method is one C-ABI call, and a failed call throws ZkError carrying the library's status
// and message (the reference's call sites `.unwrap()` their Results; an exception is the C++ form of that panic).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "fr.hpp"
#include "zkhip.h"

namespace zkhost {

using G1 = std::array<uint64_t, 18>;  // normalised Jacobian, as ark-ec Projective (include/zkhip.h)
using G1Vec = std::vector<G1>;

struct ZkError : std::runtime_error {
    int status;
    ZkError(int status_, const std::string &what) : std::runtime_error("zkhip error " + std::to_string(status_) + ": " + what), status(status_) {}
};
// `G::msm` -> Err(min_len) (dmsm.rs:23)
struct MsmLengthError : ZkError {
    using ZkError::ZkError;
    size_t min_len = 0;
};

class Ctx;

// the zk_ctx itself: shared by the Ctx object and by every buffer / SRS level / job created from it, so that the library
// context is destroyed only after the last of them (a buffer that outlives its `Ctx` variable stays valid and is freed properly)
struct CtxHandle {
    zk_ctx *h = nullptr;
    explicit CtxHandle(zk_ctx *h_) : h(h_) {}
    CtxHandle(const CtxHandle &) = delete;
    ~CtxHandle() {
        if (h) zk_ctx_destroy(h);
    }
};
using CtxRef = std::shared_ptr<CtxHandle>;

// a device allocation (zk_malloc / zk_free); DevPtr below shares it
struct DevAlloc {
    CtxRef ctx;
    void *ptr;
    size_t bytes;
    DevAlloc(CtxRef c, void *p, size_t b) : ctx(std::move(c)), ptr(p), bytes(b) {}
    DevAlloc(const DevAlloc &) = delete;
    ~DevAlloc() {
        if (ptr) zk_free(ctx->h, ptr);
    }
};

// a device address that keeps its allocation alive (a `Vec<Fr>` in HBM, or a slice of one)
struct DevPtr {
    std::shared_ptr<DevAlloc> owner;
    char *p = nullptr;
    DevPtr() {}
    DevPtr(std::shared_ptr<DevAlloc> o, char *p_) : owner(std::move(o)), p(p_) {}
    DevPtr at(size_t byte_off) const { return DevPtr(owner, p + byte_off); }
    DevPtr fr(size_t index) const { return at(32 * index); }
    void *get() const { return p; }
    explicit operator bool() const { return p != nullptr; }
};

// one SRS level resident in HBM (zk_srs)
class Srs {
  public:
    Srs(CtxRef c, zk_srs *h) : ctx_(std::move(c)), h_(h) {}
    Srs(const Srs &) = delete;
    ~Srs() {
        if (h_) zk_srs_free(ctx_->h, h_);
    }
    zk_srs *handle() const { return h_; }
    size_t len() const { return zk_srs_len(h_); }

  private:
    CtxRef ctx_;
    zk_srs *h_;
};
using SrsPtr = std::shared_ptr<Srs>;

// one item of zk_sumcheck_batch and what it returned
struct ScRequest {
    enum Kind { Plain = 0, Product = 1, Fold = 2, Open = 3 } kind;
    DevPtr f, g;
    size_t len;
    FrVec chal;  // the rounds' challenges (Plain / Product / Open: log2 len of them; Fold: any number)
    DevPtr out;  // optional: the output buffer of a Fold / Open request (allocated by the batch when empty)
};
struct ScResult {
    FrVec sums;       // Plain: 2 log2(len); Product: 3 log2(len)
    Fr last_f, last_g;  // Plain: last element; Product: both; Open: last_f = the value
    DevPtr out;       // Fold: folded table; Open: the len - 1 quotient elements
};

// test / self-check hook: the operands of one product sumcheck (zkhost/verify.hpp anchors every transcript of a run on them)
struct ScTrace {
    char kind;  // 'p' sumcheck_product, 'c' c_sumcheck_product, 'd' d_sumcheck_product
    DevPtr f, g;
    size_t len;
    FrVec challenge;
};

class Ctx {
  public:
    // when set, dist_primitive.hpp / pipeline.hpp append the operands of every product sumcheck they run (the DevPtrs keep the
    // tables alive); never read by the compute path
    std::vector<ScTrace> *sc_trace = nullptr;

    explicit Ctx(int device = 0) {
        int rc = zk_ctx_create(device, &h_);
        if (rc) throw ZkError(rc, "zk_ctx_create failed (no gfx950 device / library without device code?)");
        ref_ = std::make_shared<CtxHandle>(h_);
    }
    Ctx(const Ctx &) = delete;
    zk_ctx *handle() const { return h_; }
    void check(int rc) const {
        if (rc == ZK_ERR_LENGTH) throw MsmLengthError(rc, zk_last_error(h_));
        if (rc) throw ZkError(rc, zk_last_error(h_));
    }
    void sync() { check(zk_ctx_sync(h_)); }

    // ---- memory ----
    DevPtr alloc(size_t bytes) {
        void *p = nullptr;
        check(zk_malloc(h_, bytes ? bytes : 1, &p));
        return DevPtr(std::make_shared<DevAlloc>(ref_, p, bytes), (char *)p);
    }
    DevPtr alloc_fr(size_t n) { return alloc(32 * n); }
    void upload(const DevPtr &dst, const void *src, size_t bytes) {
        if (bytes) check(zk_memcpy_h2d(h_, dst.get(), src, bytes));
    }
    void download(void *dst, const DevPtr &src, size_t bytes) {
        if (bytes) check(zk_memcpy_d2h(h_, dst, src.get(), bytes));
    }
    void copy_d2d(const DevPtr &dst, const DevPtr &src, size_t bytes) {
        if (bytes) check(zk_memcpy_d2d(h_, dst.get(), src.get(), bytes));
    }
    DevPtr to_device(const FrVec &v) {
        DevPtr d = alloc_fr(v.size());
        upload(d, v.data(), 32 * v.size());
        return d;
    }
    FrVec to_host(const DevPtr &d, size_t n) {
        FrVec v(n);
        download(v.data(), d, 32 * n);
        return v;
    }

    // ---- element-wise Fr (dhyperplonk.rs:233-238,251-256,326-339) ----
    DevPtr fr_add(const DevPtr &a, const DevPtr &b, size_t n) { return binary(zk_fr_add, a, b, n); }
    DevPtr fr_sub(const DevPtr &a, const DevPtr &b, size_t n) { return binary(zk_fr_sub, a, b, n); }
    DevPtr fr_mul(const DevPtr &a, const DevPtr &b, size_t n) { return binary(zk_fr_mul, a, b, n); }
    DevPtr fr_batch_div(const DevPtr &a, const DevPtr &b, size_t n) { return binary(zk_fr_batch_div, a, b, n); }
    // out = a + alpha b + beta (a may be null: alpha b + beta)
    DevPtr fr_axpb(const DevPtr &a, const DevPtr &b, const Fr &alpha, const Fr &beta, size_t n) {
        DevPtr out = alloc_fr(n);
        check(zk_fr_axpb(h_, a.get(), b.get(), alpha.v, beta.v, out.get(), n));
        return out;
    }
    DevPtr fr_scale(const DevPtr &b, const Fr &alpha, size_t n) { return fr_axpb(DevPtr(), b, alpha, Fr::zero(), n); }
    // out[j osv + r osr] = sum_c M[r][c] in[j isv + c isc]   (zk_fr_apply_matrix)
    DevPtr fr_apply_matrix(const std::vector<FrVec> &m, const DevPtr &in, size_t isv, size_t isc, size_t k, size_t osv, size_t osr) {
        size_t rows = m.size(), cols = rows ? m[0].size() : 0;
        FrVec flat;
        for (auto &r : m) flat.insert(flat.end(), r.begin(), r.end());
        size_t span = (k && rows) ? (k - 1) * osv + (rows - 1) * osr + 1 : 1;
        DevPtr out = alloc_fr(span);
        check(zk_fr_apply_matrix(h_, flat.empty() ? nullptr : flat[0].v, rows, cols, in.get(), isv, isc, out.get(), osv, osr, k));
        return out;
    }
    template <class Tables>
    DevPtr fr_ntt_map(const Tables &t, const DevPtr &in, size_t isv, size_t isc, size_t k, size_t osv, size_t osr) {
        size_t span = k ? (k - 1) * osv + (t.take - 1) * osr + 1 : 1;
        DevPtr out = alloc_fr(span);
        check(zk_fr_ntt_map(h_, t.A, t.winv[0].v, t.B, t.w[0].v, t.scale[0].v, t.n_in, t.take, t.step, in.get(), isv, isc, out.get(), osv, osr, k));
        return out;
    }
    std::pair<DevPtr, DevPtr> fr_deinterleave(const DevPtr &t, size_t n) {
        DevPtr even = alloc_fr(n), odd = alloc_fr(n);
        check(zk_fr_deinterleave(h_, t.get(), even.get(), odd.get(), n));
        return {even, odd};
    }

    // ---- sumcheck family: the Phase-1 loops (dsumcheck.rs:10-21,37-85; mle.rs:88-105; dpoly_comm.rs:309-323; dacc_product.rs:31-38) ----
    ScResult sumcheck(const DevPtr &tab, size_t len, const FrVec &chal) {
        size_t n = log2_exact(len);
        need(chal.size() >= n, "sumcheck: fewer challenges than rounds");
        ScResult r;
        r.sums.resize(2 * n);
        check(zk_sumcheck(h_, tab.get(), len, n ? chal[0].v : nullptr, n ? r.sums[0].v : nullptr, r.last_f.v));
        return r;
    }
    ScResult sumcheck_product(const DevPtr &f, const DevPtr &g, size_t len, const FrVec &chal) {
        size_t n = log2_exact(len);
        need(chal.size() >= n, "sumcheck_product: fewer challenges than rounds");
        ScResult r;
        r.sums.resize(3 * n);
        check(zk_sumcheck_product(h_, f.get(), g.get(), len, n ? chal[0].v : nullptr, n ? r.sums[0].v : nullptr, r.last_f.v, r.last_g.v));
        return r;
    }
    DevPtr fold(const DevPtr &tab, size_t len, const FrVec &points) {
        size_t rounds = std::min(log2_exact(len), points.size());
        DevPtr out = alloc_fr(len >> rounds);
        check(zk_fold(h_, tab.get(), len, points.empty() ? nullptr : points[0].v, points.size(), out.get()));
        return out;
    }
    ScResult open_rounds(const DevPtr &tab, size_t len, const FrVec &point) {
        need(point.size() >= log2_exact(len), "open: fewer point coordinates than rounds");
        ScResult r;
        r.out = alloc_fr(len - 1);
        check(zk_open_rounds(h_, tab.get(), len, point.empty() ? nullptr : point[0].v, r.out.get(), r.last_f.v));
        return r;
    }
    DevPtr product_tree(const DevPtr &x, size_t N) {
        DevPtr out = alloc_fr(2 * N);
        check(zk_product_tree(h_, x.get(), N, out.get()));
        return out;
    }
    // several independent calls in one go (zk_sumcheck_batch); every output equals the single call's
    std::vector<ScResult> sumcheck_batch(const std::vector<ScRequest> &reqs) {
        std::vector<ScResult> res(reqs.size());
        std::vector<zk_sc_item> items(reqs.size());
        for (size_t i = 0; i < reqs.size(); ++i) {
            const ScRequest &q = reqs[i];
            size_t n = log2_exact(q.len);
            zk_sc_item &it = items[i];
            it = zk_sc_item{};
            it.mode = (int)q.kind, it.d_f = q.f.get(), it.d_g = q.g.get(), it.len = q.len;
            it.h_chal = q.chal.empty() ? nullptr : q.chal[0].v;
            it.n_points = q.chal.size();
            if (q.kind != ScRequest::Fold) need(q.chal.size() >= n, "sumcheck_batch: fewer challenges than rounds");
            if (q.kind == ScRequest::Plain || q.kind == ScRequest::Product) {
                res[i].sums.resize((q.kind == ScRequest::Plain ? 2 : 3) * n);
                it.h_sums = n ? res[i].sums[0].v : nullptr;
            } else if (q.kind == ScRequest::Fold) {
                res[i].out = q.out ? q.out : alloc_fr(q.len >> std::min(n, q.chal.size()));
            } else {
                res[i].out = q.out ? q.out : alloc_fr(q.len - 1);
            }
            it.h_last_f = res[i].last_f.v, it.h_last_g = res[i].last_g.v, it.d_out = res[i].out.get();
        }
        if (!reqs.empty()) check(zk_sumcheck_batch(h_, items.size(), items.data()));
        return res;
    }

    // ---- SRS ----
    SrsPtr srs_register(const void *bases, size_t stride, size_t n) {
        zk_srs *s = nullptr;
        check(zk_srs_register(h_, bases, stride, n, &s));
        return std::make_shared<Srs>(ref_, s);
    }
    // P_i = (k0 + i k1) G (canonical scalars)
    SrsPtr srs_generate(uint64_t k0, uint64_t k1, size_t n) {
        uint64_t a[4] = {k0, 0, 0, 0}, b[4] = {k1, 0, 0, 0};
        zk_srs *s = nullptr;
        check(zk_srs_generate(h_, a, b, n, &s));
        return std::make_shared<Srs>(ref_, s);
    }
    // PolynomialCommitmentCub::new (dpoly_comm.rs:37-67): levels 0 .. nvars
    std::vector<SrsPtr> srs_powers(const FrVec &s, const void *g96 = nullptr) {
        std::vector<zk_srs *> lv(s.size() + 1, nullptr);
        check(zk_srs_powers(h_, g96, s.empty() ? nullptr : s[0].v, s.size(), lv.data()));
        std::vector<SrsPtr> out;
        for (zk_srs *x : lv) out.push_back(std::make_shared<Srs>(ref_, x));
        return out;
    }
    // to_packed for ONE party (dpoly_comm.rs:164-194): row = l canonical pack coefficients
    SrsPtr srs_to_packed(const Srs &level, const FrVec &row_canonical, size_t l) {
        zk_srs *s = nullptr;
        check(zk_srs_to_packed(h_, level.handle(), row_canonical[0].v, l, &s));
        return std::make_shared<Srs>(ref_, s);
    }
    void srs_precompute(Srs &s, int window_bits = 0, int record_bytes = 0) {  // record_bytes: 0 / 96 packed (default), 128 = one G1 record per cache line
        check(record_bytes ? zk_srs_precompute_layout(h_, s.handle(), window_bits, record_bytes) : zk_srs_precompute(h_, s.handle(), window_bits));
    }

    // ---- MSM ----
    G1 msm_g1(const Srs &srs, const DevPtr &scalars, size_t n, size_t offset = 0) {
        G1 out;
        check(zk_msm_g1(h_, srs.handle(), offset, scalars.get(), n, out.data()));
        return out;
    }
    // drop-in for `G::msm(&[Affine], &[Fr]) -> Result<G, usize>` on host slices (dmsm.rs:23): bases at `stride` bytes (96, or 104 =
    // the Rust struct); a length mismatch throws MsmLengthError whose min_len is the reference's Err(min_len)
    G1 msm_g1_host(const void *bases, size_t stride, size_t n_bases, const FrVec &scalars) {
        G1 out;
        size_t min_len = 0;
        int rc = zk_msm_g1_host(h_, bases, stride, n_bases, scalars.empty() ? nullptr : scalars[0].v, scalars.size(), out.data(), &min_len);
        if (rc == ZK_ERR_LENGTH) {
            MsmLengthError e(rc, zk_last_error(h_));
            e.min_len = min_len;
            throw e;
        }
        check(rc);
        return out;
    }
    G1Vec msm_g1_batch(const std::vector<const Srs *> &srs, const std::vector<DevPtr> &scalars, const std::vector<size_t> &lens) {
        size_t count = lens.size();
        need(srs.size() == count && scalars.size() == count, "msm batch: list lengths differ");
        G1Vec out(count);
        if (!count) return out;
        std::vector<const zk_srs *> h(count);
        std::vector<const void *> sp(count);
        for (size_t i = 0; i < count; ++i) h[i] = srs[i]->handle(), sp[i] = scalars[i].get();
        check(zk_msm_g1_batch(h_, count, h.data(), nullptr, sp.data(), lens.data(), out[0].data()));
        return out;
    }
    // asynchronous form: the pass runs on the ctx's job lanes while the caller enqueues other work; wait() collects the points.
    // The scalar buffers must stay alive and unmodified until then (MsmJob holds them).
    struct MsmJob {
        CtxRef ctx;
        zk_msm_job *job = nullptr;
        size_t count = 0;
        std::vector<DevPtr> keep;
        MsmJob() {}
        MsmJob(const MsmJob &) = delete;
        MsmJob(MsmJob &&o) noexcept { *this = std::move(o); }
        MsmJob &operator=(MsmJob &&o) noexcept {
            drain();
            ctx = std::move(o.ctx), job = o.job, count = o.count, keep = std::move(o.keep);
            o.job = nullptr;
            return *this;
        }
        ~MsmJob() { drain(); }
        bool pending() const { return job != nullptr; }
        void drain() {  // never waited for: let it finish and release it
            if (job) {
                G1Vec tmp(count);
                zk_msm_wait(ctx->h, job, tmp[0].data());
                job = nullptr;
            }
        }
    };
    MsmJob msm_g1_batch_async(const std::vector<const Srs *> &srs, const std::vector<DevPtr> &scalars, const std::vector<size_t> &lens) {
        size_t count = lens.size();
        need(srs.size() == count && scalars.size() == count && count > 0, "msm batch: list lengths differ / empty");
        std::vector<const zk_srs *> h(count);
        std::vector<const void *> sp(count);
        for (size_t i = 0; i < count; ++i) h[i] = srs[i]->handle(), sp[i] = scalars[i].get();
        MsmJob j;
        j.ctx = ref_, j.count = count, j.keep = scalars;
        check(zk_msm_g1_batch_async(h_, count, h.data(), nullptr, sp.data(), lens.data(), &j.job));
        return j;
    }
    G1Vec msm_wait(MsmJob &j) {
        G1Vec out(j.count);
        zk_msm_job *job = j.job;
        j.job = nullptr;  // (zk_msm_wait releases the job also on error)
        int rc = job ? zk_msm_wait(h_, job, out[0].data()) : 0;
        j.keep.clear();  // only now: the pass read the scalar buffers until the wait returned
        check(rc);
        return out;
    }
    // the whole of d_msm in one call over the ctx's communicator (zk_d_msm)
    G1Vec d_msm(const std::vector<const Srs *> &srs, const std::vector<DevPtr> &scalars, const std::vector<size_t> &lens, const Fr *lambda_mont,
                const FrVec &coeffs_canonical) {
        size_t count = lens.size();
        G1Vec out(count);
        if (!count) return out;
        std::vector<const zk_srs *> h(count);
        std::vector<const void *> sp(count);
        for (size_t i = 0; i < count; ++i) h[i] = srs[i]->handle(), sp[i] = scalars[i].get();
        check(zk_d_msm(h_, count, h.data(), nullptr, sp.data(), lens.data(), lambda_mont ? lambda_mont->v : nullptr, coeffs_canonical[0].v, out[0].data()));
        return out;
    }
    // the level's points in the reference layout (96 B each: x || y Montgomery limbs, zeros = infinity), on the host
    std::vector<uint8_t> srs_download(const Srs &s) {
        std::vector<uint8_t> out(96 * s.len());
        if (!out.empty()) check(zk_srs_download(h_, s.handle(), out.data()));
        return out;
    }
    // zk_fr_apply_matrix on affine G1 points in HBM (96-B records): the PSS maps are generic over DomainCoeff (pss.rs:93-171).
    // m: CANONICAL scalars; out[j osv + r osr] = sum_c m[r][c] in[j isv + c isc]
    DevPtr g1_apply_matrix(const std::vector<FrVec> &m_canonical, const DevPtr &in96, size_t isv, size_t isc, size_t k, size_t osv, size_t osr) {
        size_t rows = m_canonical.size(), cols = rows ? m_canonical[0].size() : 0;
        FrVec flat;
        for (auto &r : m_canonical) flat.insert(flat.end(), r.begin(), r.end());
        size_t span = (k && rows) ? (k - 1) * osv + (rows - 1) * osr + 1 : 1;
        DevPtr out = alloc(96 * span);
        check(zk_g1_apply_matrix(h_, flat.empty() ? nullptr : flat[0].v, rows, cols, in96.get(), isv, isc, out.get(), osv, osr, k));
        return out;
    }
    // out[r] = sum_i k_i P[r n + i]; k canonical (zk_g1_lincomb_batch: the leader's public maps on points)
    G1Vec g1_lincomb_batch(const G1Vec &points, const FrVec &scalars_canonical, size_t count) {
        size_t n = scalars_canonical.size();
        need(points.size() == n * count, "g1_lincomb_batch: points != count x scalars");
        G1Vec out(count);
        if (count) check(zk_g1_lincomb_batch(h_, points[0].data(), scalars_canonical[0].v, n, count, out[0].data()));
        return out;
    }

    static size_t log2_exact(size_t len) {
        if (!len || (len & (len - 1))) throw ZkError(ZK_ERR_INVALID, "length is not a power of two");
        size_t n = 0;
        while ((size_t(1) << n) < len) ++n;
        return n;
    }

  private:
    static void need(bool ok, const char *what) {
        if (!ok) throw ZkError(ZK_ERR_INVALID, what);
    }
    template <class F>
    DevPtr binary(F fn, const DevPtr &a, const DevPtr &b, size_t n) {
        DevPtr out = alloc_fr(n);
        check(fn(h_, a.get(), b.get(), out.get(), n));
        return out;
    }
    zk_ctx *h_ = nullptr;
    CtxRef ref_;
};

}  // namespace zkhost
