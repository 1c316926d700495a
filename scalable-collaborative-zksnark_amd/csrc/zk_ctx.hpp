// zk_ctx.hpp -- per-GPU context shared by the libzkhip translation units (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/zkhip.h"

struct zk_srs {
    void* d_bases = nullptr;  // packed 96-B affine points (x||y Montgomery, x=y=0: infinity)
    size_t n = 0;
    bool owned = true;
    bool g2 = false;  // points of G2 (192-byte affine records over Fq2, no endomorphism copies)
    // optional precomputed table: copy w of point i = 2^{bit_offset(w)} * P_i at d_table[w * table_stride + i]
    void* d_table = nullptr;
    int table_c = 0;
    size_t table_stride = 0;
};

struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;
    // growable device scratch arenas (never shrunk; freed with the ctx)
    struct Arena {
        void* p = nullptr;
        size_t cap = 0;
    };
    Arena scratch[12];
    void* h_pinned = nullptr;  // pinned host staging
    size_t h_pinned_cap = 0;
    int msm_window_override = 0;
    float msm_ms[6] = {0, 0, 0, 0, 0, 0};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    static constexpr int kAux = 6;  // extra streams for independent MSM window classes of one batch
    hipStream_t aux[kAux] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kAux] = {};
    static constexpr int kParts = 4;  // staggered parts of one MSM class (zk_msm.hip)
    hipEvent_t ev_part[kParts] = {};
    std::vector<hipEvent_t> ev_cls;  // completion event per MSM window class of a batch
    void* host_pool = nullptr;       // worker threads for the per-item host chains (zk_msm.hip)
    // zk_malloc / zk_free block recycling (zk_api.cpp)
    std::unordered_map<size_t, std::vector<void*>> pool_free;
    std::unordered_map<void*, size_t> pool_size;
    size_t pool_bytes = 0;
    std::mutex pool_mu;  // a garbage collector may release buffers from another thread
    int cu_count = 256;
    // party exchanges (zk_comm.cpp): an RCCL communicator bound to this ctx's GPU
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
};

namespace zk {

int fail(zk_ctx* ctx, int code, const char* fmt, ...);
int hip_fail(zk_ctx* ctx, hipError_t e, const char* what);
// returns device scratch of at least `bytes` (slot 0..7), or nullptr after recording the error
void* scratch(zk_ctx* ctx, int slot, size_t bytes);
void* pinned(zk_ctx* ctx, size_t bytes);
// hipMalloc that drops the ctx's parked zk_free blocks and retries once on out-of-memory
hipError_t device_alloc(zk_ctx* ctx, void** out, size_t bytes, bool pool_locked = false);
size_t pool_trim(zk_ctx* ctx);

#define ZK_HIP(ctx, call)                                            \
    do {                                                             \
        hipError_t _e = (call);                                      \
        if (_e != hipSuccess) return zk::hip_fail(ctx, _e, #call);   \
    } while (0)

// ---- zk_fr.hip ----
int fr_binary(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n);
int fr_axpb(zk_ctx* ctx, const void* a, const void* b, const uint64_t* alpha, const uint64_t* beta, void* out, size_t n);
int fr_apply_matrix(zk_ctx* ctx, const uint64_t* h_matrix, size_t rows, size_t cols, const void* d_in, size_t isv, size_t isc,
                    void* d_out, size_t osv, size_t osr, size_t k);
int fr_ntt_map(zk_ctx* ctx, size_t A, const uint64_t* h_winv, size_t B, const uint64_t* h_w, const uint64_t* h_scale, size_t nin, size_t take,
               size_t step, const void* d_in, size_t isv, size_t isc, void* d_out, size_t osv, size_t osr, size_t k);
int fr_deinterleave(zk_ctx* ctx, const void* t, void* even, void* odd, size_t n);
int fr_batch_div(zk_ctx* ctx, const void* num, const void* den, void* out, size_t n);
// mode 0 plain sums, 1 product sums, 2 fold only, 3 open quotients
int multilinear_run(zk_ctx* ctx, int mode, const void* d_f, const void* d_g, size_t len, const uint64_t* h_chal,
                    size_t rounds, uint64_t* h_sums, uint64_t* h_last_f, uint64_t* h_last_g, void* d_out, void* d_q);
int product_tree(zk_ctx* ctx, const void* d_x, size_t N, void* d_tree);
int dbg_fq(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n);

// ---- zk_msm.hip ----
struct MsmItem {
    const zk_srs* srs;
    size_t offset;
    const void* d_scalars;
    size_t n;
};
int msm_g1_batch(zk_ctx* ctx, const MsmItem* items, size_t count, uint64_t* h_out);
int msm_g2_batch(zk_ctx* ctx, const MsmItem* items, size_t count, uint64_t* h_out);  // h_out: 36 u64 per item
int srs_pack_g2(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out);
int msm_g1(zk_ctx* ctx, const zk_srs* srs, size_t offset, const void* d_scalars, size_t n, uint64_t* h_out);
int srs_pack(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out);
int srs_precompute(zk_ctx* ctx, zk_srs* srs, int c);
void msm_host_pool_destroy(zk_ctx* ctx);
int srs_from_device(zk_ctx* ctx, const void* d_bases96, size_t n, zk_srs** out);
int srs_download(zk_ctx* ctx, const zk_srs* srs, void* h_out96);
int srs_generate(zk_ctx* ctx, const uint64_t* k0, const uint64_t* k1, size_t n, zk_srs** out);
// ---- zk_srs.hip ----
int srs_powers(zk_ctx* ctx, const void* h_g96, const uint64_t* h_s, size_t nvars, zk_srs** out_levels);
int srs_to_packed(zk_ctx* ctx, const zk_srs* level, const uint64_t* h_row, size_t l, zk_srs** out);
int g1_apply_matrix_ref(zk_ctx* ctx, const uint64_t* h_matrix, size_t rows, size_t cols, const void* d_in, size_t isv, size_t isc, void* d_out,
                        size_t osv, size_t osr, size_t k);
int dbg_g1_op(zk_ctx* ctx, int mode, const void* p, const void* q, void* h_out, size_t n);
int dbg_g2_op(zk_ctx* ctx, int mode, const void* p, const void* q, void* h_out, size_t n);
int msm_pick_window(size_t n);
int g1_lincomb_batch_host(zk_ctx* ctx, const uint64_t* h_points_jac, const uint64_t* h_scalars_canon, size_t n, size_t count,
                          uint64_t* h_out);
int g1_lincomb_host(zk_ctx* ctx, const uint64_t* h_points_jac, const uint64_t* h_scalars_canon, size_t n, uint64_t* h_out);

}  // namespace zk
