// zk_api.cpp -- extern "C" surface of libzkhip.so (see include/zkhip.h) and context plumbing.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "zk_ctx.hpp"

namespace zk {

struct TuneKey {
    const char* name;
    long Tuning::*field;
};
static const TuneKey kTuneKeys[] = {
    {"sc_pass_wg", &Tuning::sc_pass_wg}, {"sc_local_g", &Tuning::sc_local_g}, {"sc_local_threads", &Tuning::sc_local_threads}, {"sc_ts", &Tuning::sc_ts}, {"sc_xcd", &Tuning::sc_xcd},
    {"sc_kf", &Tuning::sc_kf}, {"sc_plain_flat", &Tuning::sc_plain_flat}, {"sc_kp", &Tuning::sc_kp}, {"sc_k0", &Tuning::sc_k0},
    {"sc_flat_wg", &Tuning::sc_flat_wg}, {"sc_plain_wg", &Tuning::sc_plain_wg}, {"sc_pre", &Tuning::sc_pre},
    {"sc_pinned_out", &Tuning::sc_pinned_out}, {"sc_t1_device", &Tuning::sc_t1_device}, {"sc_handover", &Tuning::sc_handover},
    {"msm_table_dc", &Tuning::msm_table_dc}, {"msm_qstep", &Tuning::msm_qstep}, {"msm_tile", &Tuning::msm_tile}, {"msm_pair", &Tuning::msm_pair},
    {"msm_fixq", &Tuning::msm_fixq}, {"msm_quad", &Tuning::msm_quad}, {"msm_stage", &Tuning::msm_stage}, {"msm_split", &Tuning::msm_split},
    {"msm_np", &Tuning::msm_np}, {"msm_fused_min", &Tuning::msm_fused_min}, {"msm_l2_tiled", &Tuning::msm_l2_tiled}, {"msm_tab_spt", &Tuning::msm_tab_spt}, {"msm_idx_ahead", &Tuning::msm_idx_ahead}, {"msm_share_l1", &Tuning::msm_share_l1}, {"srs_table_batched", &Tuning::srs_table_batched}, {"msm_debug", &Tuning::msm_debug}, {"msm_serial", &Tuning::msm_serial}, {"msm_size_classes", &Tuning::msm_size_classes},
    {"g1_map_by_column", &Tuning::g1_map_by_column}, {"msm_share", &Tuning::msm_share}, {"srs_table_rec", &Tuning::srs_table_rec}, {"msm_size_class_min", &Tuning::msm_size_class_min}, {"msm_small_table_widths", &Tuning::msm_small_table_widths},
};
static Tuning g_tuning;
int tune_set(const char* key, long value) {
    if (!key) return ZK_ERR_INVALID;
    for (const TuneKey& k : kTuneKeys)
        if (!std::strcmp(k.name, key)) {
            tuning().*(k.field) = value;
            return ZK_OK;
        }
    return ZK_ERR_INVALID;
}
// the ONE environment read of the library: ZKHIP_TUNE="key=value,key=value" (unknown keys are reported on stderr)
Tuning& tuning() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = std::getenv("ZKHIP_TUNE");
        if (!e) return;
        std::string s(e);
        size_t pos = 0;
        while (pos < s.size()) {
            size_t end = s.find(',', pos);
            if (end == std::string::npos) end = s.size();
            const std::string kv = s.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            bool ok = false;
            if (eq != std::string::npos) {
                const std::string k = kv.substr(0, eq);
                for (const TuneKey& t : kTuneKeys)
                    if (k == t.name) {
                        g_tuning.*(t.field) = std::atol(kv.c_str() + eq + 1);
                        ok = true;
                    }
            }
            if (!ok && !kv.empty()) fprintf(stderr, "zkhip: ZKHIP_TUNE: unknown entry '%s'\n", kv.c_str());
            pos = end + 1;
        }
    });
    return g_tuning;
}

int fail(zk_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}
int hip_fail(zk_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, e == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
// releases every parked zk_free block of the ctx (pool_mu held by the caller)
static size_t pool_flush_locked(zk_ctx* ctx) {
    size_t freed = ctx->pool_bytes;
    if (!ctx->pool_free.empty()) hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pool_free)
        for (void* p : kv.second) hipFree(p);
    ctx->pool_free.clear();
    ctx->pool_bytes = 0;
    return freed;
}
size_t pool_trim(zk_ctx* ctx) {
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    return pool_flush_locked(ctx);
}
// the ONE hipMalloc site of the library: on out-of-memory the parked blocks are dropped and the
// allocation retried once (the park is invisible to hipMalloc, torch and RCCL otherwise)
// Large requests leave kRuntimeReserve of the device to the runtime: a kernel dispatch needs memory of its own (the private segments of
// the big accumulation kernels, signals, kernarg pools), and a device filled to the last GiB by arenas does not fail a hipMalloc -- it
// fails the NEXT DISPATCH with HSA_STATUS_ERROR_OUT_OF_RESOURCES and a queue dump, which kills the process (seen with 8 parties' arenas on
// one GPU).  Refusing the allocation instead is ZK_ERR_OOM: an error the caller can act on.
static constexpr size_t kRuntimeReserve = (size_t)2 << 30;
static bool leaves_reserve(size_t bytes) {
    if (bytes < ((size_t)64 << 20)) return true;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return true;
    return fr >= bytes && fr - bytes >= kRuntimeReserve;
}
hipError_t device_alloc(zk_ctx* ctx, void** out, size_t bytes, bool pool_locked) {
    hipError_t e = leaves_reserve(bytes) ? hipMalloc(out, bytes) : hipErrorOutOfMemory;
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    if (pool_locked) pool_flush_locked(ctx);
    else pool_trim(ctx);
    if (!leaves_reserve(bytes)) return hipErrorOutOfMemory;
    return hipMalloc(out, bytes);
}
void* scratch(zk_ctx* ctx, int slot, size_t bytes) {
    zk_ctx::Arena& a = ctx->scratch[slot];
    if (bytes <= a.cap && a.p) return a.p;
    if (a.p) {
        hipStreamSynchronize(ctx->stream);
        hipFree(a.p);
        a.p = nullptr;
        a.cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = device_alloc(ctx, &a.p, want);
    if (e != hipSuccess) {
        hip_fail(ctx, e, "hipMalloc(scratch)");
        a.p = nullptr;
        return nullptr;
    }
    a.cap = want;
    return a.p;
}
void* pinned(zk_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->h_pinned_cap && ctx->h_pinned) return ctx->h_pinned;
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    ctx->h_pinned = nullptr;
    size_t want = bytes < 4096 ? 4096 : bytes;
    hipError_t e = hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        hip_fail(ctx, e, "hipHostMalloc");
        ctx->h_pinned = nullptr;
        ctx->h_pinned_cap = 0;
        return nullptr;
    }
    ctx->h_pinned_cap = want;
    return ctx->h_pinned;
}

}  // namespace zk

using namespace zk;

extern "C" {

const char* zk_version(void) { return "zkhip 0.1 (gfx950)"; }

// ---- the arena plan (include/zkhip.h): [magic, lanes, 12 scratch caps, pinned cap, 3 x (10 lane caps, lane pinned cap)] ----
static constexpr uint64_t kPlanMagic = 0x31304e414c504b5aull;  // "ZKPLAN01"
int zk_arena_plan_export(zk_ctx* ctx, uint64_t* h_plan) {
    if (!ctx || !h_plan) return ZK_ERR_INVALID;
    static_assert(2 + 12 + 1 + (zk_ctx::kLanes - 1) * 11 == ZK_ARENA_PLAN_WORDS, "plan layout");
    size_t k = 0;
    h_plan[k++] = kPlanMagic;
    h_plan[k++] = zk_ctx::kLanes;
    for (int i = 0; i < 12; i++) h_plan[k++] = ctx->scratch[i].p ? ctx->scratch[i].cap : 0;
    h_plan[k++] = ctx->h_pinned ? ctx->h_pinned_cap : 0;
    for (int l = 1; l < zk_ctx::kLanes; l++) {
        for (int i = 0; i < 10; i++) h_plan[k++] = ctx->lanes[l].mem[i].p ? ctx->lanes[l].mem[i].cap : 0;
        h_plan[k++] = ctx->lanes[l].pinned ? ctx->lanes[l].pinned_cap : 0;
    }
    return ZK_OK;
}
int zk_arena_plan_import(zk_ctx* ctx, const uint64_t* h_plan) {
    if (!ctx || !h_plan) return ZK_ERR_INVALID;
    if (h_plan[0] != kPlanMagic || h_plan[1] != (uint64_t)zk_ctx::kLanes) return fail(ctx, ZK_ERR_INVALID, "zk_arena_plan_import: not an arena plan of this library version");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    size_t k = 2;
    for (int i = 0; i < 12; i++, k++)
        if (h_plan[k] && !scratch(ctx, i, (size_t)h_plan[k])) return ZK_ERR_OOM;
    if (h_plan[k] && !pinned(ctx, (size_t)h_plan[k])) return ZK_ERR_OOM;
    k++;
    for (int l = 1; l < zk_ctx::kLanes; l++, k += 11) {
        const int rc = msm_lanes_reserve(ctx, l, h_plan + k, h_plan[k + 10]);
        if (rc) return rc;
    }
    return ZK_OK;
}

int zk_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return ZK_ERR_NO_DEVICE;
    return count;
}

int zk_ctx_create(int device_id, zk_ctx** out) {
    if (!out) return ZK_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return ZK_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return ZK_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return ZK_ERR_HIP;
    zk_ctx* c = new zk_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ZK_ERR_HIP;
    }
    c->own_stream = true;
    for (auto& e : c->ev) hipEventCreate(&e);
    {  // lane 0 of the MSM pipeline = the ctx stream + auxiliary streams (the async lanes are created on first use)
        zk_ctx::MsmLane& L = c->lanes[0];
        L.main = c->stream;
        for (auto& s2 : L.aux) hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming);
        for (auto& e : L.ev_join) hipEventCreateWithFlags(&e, hipEventDisableTiming);
        for (auto& e : L.ev_part) hipEventCreateWithFlags(&e, hipEventDisableTiming);
        L.ready = true;
    }
    hipEventCreateWithFlags(&c->ev_async_in, hipEventDisableTiming);
    *out = c;
    return ZK_OK;
}
void zk_ctx_destroy(zk_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    zk_comm_destroy(ctx);
    for (auto& a : ctx->scratch)
        if (a.p) hipFree(a.p);
    for (auto& kv : ctx->pool_free)
        for (void* p : kv.second) hipFree(p);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    if (ctx->h_comm) hipHostFree(ctx->h_comm);
    for (auto& e : ctx->ev)
        if (e) hipEventDestroy(e);
    zk::msm_lanes_destroy(ctx);
    if (ctx->ev_async_in) hipEventDestroy(ctx->ev_async_in);
    zk::msm_host_pool_destroy(ctx);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}
const char* zk_last_error(zk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
int zk_ctx_set_stream(zk_ctx* ctx, void* hip_stream) {
    if (!ctx) return ZK_ERR_INVALID;
    hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return ZK_ERR_HIP;
        ctx->own_stream = true;
    }
    ctx->lanes[0].main = ctx->stream;
    return ZK_OK;
}
int zk_ctx_sync(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_INVALID;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// Device buffers are recycled per ctx: zk_free parks the block (exact size) instead of hipFree -- which
// would synchronise the device -- and zk_malloc hands it out again.  Safe because every use of a ctx's
// memory is ordered on the ctx stream.  The park is capped (kPoolCapBytes); beyond it blocks are really freed.
static constexpr size_t kPoolCapBytes = (size_t)32 << 30;
int zk_malloc(zk_ctx* ctx, size_t bytes, void** d_out) {
    if (!ctx || !d_out) return ZK_ERR_INVALID;
    if (!bytes) bytes = 1;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    auto it = ctx->pool_free.find(bytes);
    if (it != ctx->pool_free.end() && !it->second.empty()) {
        *d_out = it->second.back();
        it->second.pop_back();
        ctx->pool_bytes -= bytes;
        ctx->pool_size[*d_out] = bytes;
        return ZK_OK;
    }
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ZK_HIP(ctx, device_alloc(ctx, d_out, bytes, /*pool_locked=*/true));
    ctx->pool_size[*d_out] = bytes;
    return ZK_OK;
}
int zk_trim(zk_ctx* ctx, size_t* h_freed_bytes) {
    if (!ctx) return ZK_ERR_INVALID;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t freed = pool_trim(ctx);
    if (h_freed_bytes) *h_freed_bytes = freed;
    return ZK_OK;
}
int zk_free(zk_ctx* ctx, void* d_ptr) {
    if (!ctx) return ZK_ERR_INVALID;
    if (!d_ptr) return ZK_OK;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    auto it = ctx->pool_size.find(d_ptr);
    if (it != ctx->pool_size.end()) {
        const size_t bytes = it->second;
        ctx->pool_size.erase(it);
        if (ctx->pool_bytes + bytes <= kPoolCapBytes) {
            ctx->pool_free[bytes].push_back(d_ptr);
            ctx->pool_bytes += bytes;
            return ZK_OK;
        }
    }
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP(ctx, hipFree(d_ptr));
    return ZK_OK;
}
int zk_memcpy_h2d(zk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx) return ZK_ERR_INVALID;
    ZK_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
int zk_memcpy_d2h(zk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx) return ZK_ERR_INVALID;
    ZK_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

int zk_mem_info(zk_ctx* ctx, size_t* h_free, size_t* h_total) {
    if (!ctx) return ZK_ERR_INVALID;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    ZK_HIP(ctx, hipMemGetInfo(&f, &t));
    if (h_free) *h_free = f;
    if (h_total) *h_total = t;
    return ZK_OK;
}
int zk_memcpy_d2d(zk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (!ctx) return ZK_ERR_INVALID;
    if (bytes) ZK_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ZK_OK;
}

#define NEED(ctx, cond)                                              \
    do {                                                             \
        if (!(ctx)) return ZK_ERR_INVALID;                           \
        if (!(cond)) return fail(ctx, ZK_ERR_INVALID, "null argument"); \
    } while (0)

int zk_fr_add(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n) {
    NEED(ctx, n == 0 || (a && b && out));
    return fr_binary(ctx, 0, a, b, out, n);
}
int zk_fr_sub(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n) {
    NEED(ctx, n == 0 || (a && b && out));
    return fr_binary(ctx, 1, a, b, out, n);
}
int zk_fr_mul(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n) {
    NEED(ctx, n == 0 || (a && b && out));
    return fr_binary(ctx, 2, a, b, out, n);
}
int zk_fr_axpb(zk_ctx* ctx, const void* a, const void* b, const uint64_t alpha[4], const uint64_t beta[4], void* out, size_t n) {
    NEED(ctx, alpha && beta && (n == 0 || (b && out)));  // a may be NULL (treated as zero)
    return fr_axpb(ctx, a, b, alpha, beta, out, n);
}
int zk_fr_apply_matrix(zk_ctx* ctx, const uint64_t* h_matrix, size_t rows, size_t cols, const void* d_in, size_t in_vec_stride,
                       size_t in_comp_stride, void* d_out, size_t out_vec_stride, size_t out_row_stride, size_t k) {
    NEED(ctx, k == 0 || rows == 0 || (h_matrix && d_in && d_out));
    return fr_apply_matrix(ctx, h_matrix, rows, cols, d_in, in_vec_stride, in_comp_stride, d_out, out_vec_stride, out_row_stride, k);
}
int zk_fr_ntt_map(zk_ctx* ctx, size_t A, const uint64_t* h_winv, size_t B, const uint64_t* h_w, const uint64_t* h_scale, size_t n_in,
                  size_t take, size_t step, const void* d_in, size_t in_vec_stride, size_t in_comp_stride, void* d_out, size_t out_vec_stride,
                  size_t out_row_stride, size_t k) {
    NEED(ctx, k == 0 || take == 0 || (h_winv && h_w && h_scale && d_in && d_out));
    return fr_ntt_map(ctx, A, h_winv, B, h_w, h_scale, n_in, take, step, d_in, in_vec_stride, in_comp_stride, d_out, out_vec_stride, out_row_stride, k);
}
int zk_fr_deinterleave(zk_ctx* ctx, const void* d_t, void* d_even, void* d_odd, size_t n) {
    NEED(ctx, n == 0 || (d_t && d_even && d_odd));
    return fr_deinterleave(ctx, d_t, d_even, d_odd, n);
}
int zk_fr_batch_div(zk_ctx* ctx, const void* num, const void* den, void* out, size_t n) {
    NEED(ctx, n == 0 || (num && den && out));
    return fr_batch_div(ctx, num, den, out, n);
}

static int log2_exact(size_t len) {
    int n = 0;
    while (((size_t)1 << n) < len) n++;
    return n;
}
int zk_sumcheck(zk_ctx* ctx, const void* d_tab, size_t len, const uint64_t* h_chal, uint64_t* h_out_pairs, uint64_t h_last[4]) {
    NEED(ctx, d_tab && h_last && (len <= 1 || (h_chal && h_out_pairs)));
    return multilinear_run(ctx, 0, d_tab, nullptr, len, h_chal, (size_t)log2_exact(len), h_out_pairs, h_last, nullptr, nullptr, nullptr);
}
int zk_sumcheck_product(zk_ctx* ctx, const void* d_f, const void* d_g, size_t len, const uint64_t* h_chal, uint64_t* h_out_triples,
                        uint64_t h_last_f[4], uint64_t h_last_g[4]) {
    NEED(ctx, d_f && d_g && h_last_f && h_last_g && (len <= 1 || (h_chal && h_out_triples)));
    return multilinear_run(ctx, 1, d_f, d_g, len, h_chal, (size_t)log2_exact(len), h_out_triples, h_last_f, h_last_g, nullptr, nullptr);
}
int zk_fold(zk_ctx* ctx, const void* d_tab, size_t len, const uint64_t* h_points, size_t n_points, void* d_out) {
    NEED(ctx, d_tab && d_out && (n_points == 0 || h_points));
    if (len == 0 || (len & (len - 1))) return fail(ctx, ZK_ERR_INVALID, "table length %zu is not a power of two", len);
    size_t n = (size_t)log2_exact(len);
    size_t rounds = n_points < n ? n_points : n;  // min(n, points_cnt), mle.rs:94
    return multilinear_run(ctx, 2, d_tab, nullptr, len, h_points, rounds, nullptr, nullptr, nullptr, d_out, nullptr);
}
int zk_open_rounds(zk_ctx* ctx, const void* d_tab, size_t len, const uint64_t* h_point, void* d_q_out, uint64_t h_value[4]) {
    NEED(ctx, d_tab && h_value && (len <= 1 || (h_point && d_q_out)));
    return multilinear_run(ctx, 3, d_tab, nullptr, len, h_point, (size_t)log2_exact(len), nullptr, h_value, nullptr, nullptr, d_q_out);
}
int zk_sumcheck_batch(zk_ctx* ctx, size_t count, const zk_sc_item* items) {
    NEED(ctx, count == 0 || items);
    return multilinear_batch(ctx, items, count);
}
int zk_product_tree(zk_ctx* ctx, const void* d_x, size_t N, void* d_tree) {
    NEED(ctx, d_x && d_tree);
    return product_tree(ctx, d_x, N, d_tree);
}

int zk_srs_register(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out) {
    NEED(ctx, out);
    return srs_pack(ctx, h_bases, stride, n, out);
}
int zk_srs_wrap_device(zk_ctx* ctx, const void* d_bases96, size_t n, zk_srs** out) {
    NEED(ctx, out && (n == 0 || d_bases96));
    return srs_from_device(ctx, d_bases96, n, out);
}
int zk_srs_download(zk_ctx* ctx, const zk_srs* srs, void* h_out96) {
    NEED(ctx, srs);
    return srs_download(ctx, srs, h_out96);
}
int zk_srs_generate(zk_ctx* ctx, const uint64_t k0[4], const uint64_t k1[4], size_t n, zk_srs** out) {
    NEED(ctx, out && k0 && k1);
    return srs_generate(ctx, k0, k1, n, out);
}
int zk_srs_powers(zk_ctx* ctx, const void* h_g96, const uint64_t* h_s, size_t nvars, zk_srs** out_levels) {
    NEED(ctx, out_levels && (nvars == 0 || h_s));
    return srs_powers(ctx, h_g96, h_s, nvars, out_levels);
}
int zk_srs_to_packed(zk_ctx* ctx, const zk_srs* level, const uint64_t* h_row, size_t l, zk_srs** out) {
    NEED(ctx, level && h_row && out);
    return srs_to_packed(ctx, level, h_row, l, out);
}
int zk_g1_apply_matrix(zk_ctx* ctx, const uint64_t* h_matrix, size_t rows, size_t cols, const void* d_in96, size_t in_vec_stride,
                       size_t in_comp_stride, void* d_out96, size_t out_vec_stride, size_t out_row_stride, size_t k) {
    NEED(ctx, k == 0 || rows == 0 || (h_matrix && d_in96 && d_out96));
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    return g1_apply_matrix_ref(ctx, h_matrix, rows, cols, d_in96, in_vec_stride, in_comp_stride, d_out96, out_vec_stride, out_row_stride, k);
}
int zk_srs_precompute(zk_ctx* ctx, zk_srs* srs, int window_bits) {
    NEED(ctx, srs);
    return srs_precompute(ctx, srs, window_bits, 0);
}
int zk_srs_precompute_layout(zk_ctx* ctx, zk_srs* srs, int window_bits, int record_bytes) {
    NEED(ctx, srs);
    if (record_bytes != 0 && record_bytes != 96 && record_bytes != 128) return fail(ctx, ZK_ERR_INVALID, "zk_srs_precompute_layout: record_bytes must be 0 (default), 96 or 128");
    return srs_precompute(ctx, srs, window_bits, record_bytes);
}
int zk_srs_table_record(const zk_srs* srs) { return (srs && srs->d_table) ? (int)srs->table_rec : 0; }
int zk_srs_free(zk_ctx* ctx, zk_srs* srs) {
    if (!srs) return ZK_OK;
    if (srs->d_table) {
        if (ctx) hipStreamSynchronize(ctx->stream);
        hipFree(srs->d_table);
    }
    if (srs->owned && srs->d_bases) {
        if (ctx) hipStreamSynchronize(ctx->stream);
        hipFree(srs->d_bases);
    }
    delete srs;
    return ZK_OK;
}
int zk_srs_table_window(const zk_srs* srs) { return (srs && srs->d_table) ? srs->table_c : 0; }
size_t zk_srs_len(const zk_srs* srs) { return srs ? srs->n : 0; }
const void* zk_srs_device_ptr(const zk_srs* srs) { return srs ? srs->d_bases : nullptr; }

int zk_msm_g1(zk_ctx* ctx, const zk_srs* srs, size_t offset, const void* d_scalars, size_t n, uint64_t h_out[18]) {
    NEED(ctx, srs && h_out && (n == 0 || d_scalars));
    return msm_g1(ctx, srs, offset, d_scalars, n, h_out);
}
int zk_msm_g1_batch(zk_ctx* ctx, size_t count, const zk_srs* const* srs, const size_t* offsets, const void* const* d_scalars,
                    const size_t* n, uint64_t* h_out) {
    NEED(ctx, count == 0 || (srs && d_scalars && n && h_out));
    std::vector<MsmItem> items(count);
    for (size_t k = 0; k < count; k++) items[k] = MsmItem{srs[k], offsets ? offsets[k] : 0, d_scalars[k], n[k]};
    return msm_g1_batch(ctx, items.data(), count, h_out);
}
int zk_msm_g1_batch_async(zk_ctx* ctx, size_t count, const zk_srs* const* srs, const size_t* offsets, const void* const* d_scalars,
                          const size_t* n, zk_msm_job** job) {
    NEED(ctx, job && (count == 0 || (srs && d_scalars && n)));
    std::vector<MsmItem> items(count);
    for (size_t k = 0; k < count; k++) items[k] = MsmItem{srs[k], offsets ? offsets[k] : 0, d_scalars[k], n[k]};
    return msm_g1_batch_async(ctx, items.data(), count, job);
}
int zk_msm_wait(zk_ctx* ctx, zk_msm_job* job, uint64_t* h_out) {
    NEED(ctx, job);
    return msm_job_wait(ctx, job, h_out);
}
int zk_srs_register_g2(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out) {
    NEED(ctx, out);
    return srs_pack_g2(ctx, h_bases, stride, n, out);
}
int zk_msm_g2(zk_ctx* ctx, const zk_srs* srs, size_t offset, const void* d_scalars, size_t n, uint64_t h_out[36]) {
    NEED(ctx, srs && h_out && (n == 0 || d_scalars));
    MsmItem it{srs, offset, d_scalars, n};
    return msm_g2_batch(ctx, &it, 1, h_out);
}
int zk_msm_g2_batch(zk_ctx* ctx, size_t count, const zk_srs* const* srs, const size_t* offsets, const void* const* d_scalars,
                    const size_t* n, uint64_t* h_out) {
    NEED(ctx, count == 0 || (srs && d_scalars && n && h_out));
    std::vector<MsmItem> items(count);
    for (size_t k = 0; k < count; k++) items[k] = MsmItem{srs[k], offsets ? offsets[k] : 0, d_scalars[k], n[k]};
    return msm_g2_batch(ctx, items.data(), count, h_out);
}
int zk_msm_g1_host(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n_bases, const uint64_t* h_scalars, size_t n_scalars,
                   uint64_t h_out[18], size_t* h_err_len) {
    NEED(ctx, h_out);
    if (n_bases != n_scalars) {  // VariableBaseMSM::msm -> Err(min(len)) (ark-ec 0.4.2), unwrap()ed at dmsm.rs:23
        size_t m = n_bases < n_scalars ? n_bases : n_scalars;
        if (h_err_len) *h_err_len = m;
        return fail(ctx, ZK_ERR_LENGTH, "msm: bases.len() = %zu != scalars.len() = %zu (Err(%zu))", n_bases, n_scalars, m);
    }
    zk_srs* srs = nullptr;
    int rc = srs_pack(ctx, h_bases, stride, n_bases, &srs);
    if (rc) return rc;
    void* d_s = nullptr;
    if (n_scalars) {
        d_s = scratch(ctx, 7, n_scalars * 32);
        if (!d_s) {
            zk_srs_free(ctx, srs);
            return ZK_ERR_OOM;
        }
        hipError_t e = hipMemcpyAsync(d_s, h_scalars, n_scalars * 32, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            zk_srs_free(ctx, srs);
            return hip_fail(ctx, e, "hipMemcpyAsync(scalars)");
        }
    }
    rc = msm_g1(ctx, srs, 0, d_s, n_scalars, h_out);
    zk_srs_free(ctx, srs);
    return rc;
}
int zk_g1_lincomb(zk_ctx* ctx, const uint64_t* h_points, const uint64_t* h_scalars, size_t n, uint64_t h_out[18]) {
    NEED(ctx, h_out && (n == 0 || (h_points && h_scalars)));
    return g1_lincomb_host(ctx, h_points, h_scalars, n, h_out);
}
int zk_g1_lincomb_batch(zk_ctx* ctx, const uint64_t* h_points, const uint64_t* h_scalars, size_t n, size_t count, uint64_t* h_out) {
    NEED(ctx, (count == 0 || h_out) && (n == 0 || count == 0 || (h_points && h_scalars)));
    return g1_lincomb_batch_host(ctx, h_points, h_scalars, n, count, h_out);
}
int zk_msm_window(size_t n) { return msm_pick_window(n); }
int zk_msm_set_window(zk_ctx* ctx, int c) {
    if (!ctx || c < 0 || c > 20) return ZK_ERR_INVALID;
    ctx->msm_window_override = c;
    return ZK_OK;
}
int zk_msm_last_timing(zk_ctx* ctx, float h_ms[6]) {
    if (!ctx || !h_ms) return ZK_ERR_INVALID;
    std::memcpy(h_ms, ctx->msm_ms, sizeof(ctx->msm_ms));
    return ZK_OK;
}

int zk_sumcheck_last_timing(zk_ctx* ctx, float h_ms[2]) {
    if (!ctx || !h_ms) return ZK_ERR_INVALID;
    h_ms[0] = ctx->sc_ms[0], h_ms[1] = ctx->sc_ms[1];
    return ZK_OK;
}
int zk_dbg_tune(const char* key, long value) { return tune_set(key, value); }

int zk_dbg_fq_mul2add(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n) { return dbg_fq(ctx, 3, a, b, out, n); }
int zk_dbg_fq_add(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n) { return dbg_fq(ctx, 0, a, b, out, n); }
int zk_dbg_fq_sub(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n) { return dbg_fq(ctx, 1, a, b, out, n); }
int zk_dbg_fq_mul(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n) { return dbg_fq(ctx, 2, a, b, out, n); }
int zk_dbg_g1_op(zk_ctx* ctx, int mode, const void* p, const void* q, void* h_out, size_t n) {
    NEED(ctx, n == 0 || (p && q && h_out));
    return dbg_g1_op(ctx, mode, p, q, h_out, n);
}

int zk_dbg_g2_op(zk_ctx* ctx, int mode, const void* p, const void* q, void* h_out, size_t n) {
    NEED(ctx, n == 0 || (p && q && h_out));
    return dbg_g2_op(ctx, mode, p, q, h_out, n);
}

}  // extern "C"
