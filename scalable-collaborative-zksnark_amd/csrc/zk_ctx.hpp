// zk_ctx.hpp -- per-GPU context shared by the libzkhip translation units (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/zkhip.h"
#include "../../include/zkhip_test.h"  // (the zk_dbg_* hooks are defined in this library too)

struct zk_msm_job;  // zk_msm.hip

struct zk_srs {
    void* d_bases = nullptr;  // packed 96-B affine points (x||y Montgomery, x=y=0: infinity)
    size_t n = 0;
    bool owned = true;
    bool g2 = false;  // points of G2 (192-byte affine records over Fq2, no endomorphism copies)
    // optional precomputed table: copy w of point i = 2^{bit_offset(w)} * P_i at d_table[w * table_stride + i]
    void* d_table = nullptr;
    int table_c = 0;
    size_t table_stride = 0;
    size_t table_rec = 96;  // bytes per record of d_table (G1: 96 packed, or 128 = one record per line; G2: 192)
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
    float sc_ms[2] = {0, 0};  // tuning sc_ts = 3: device time of the first stage / of all launches of the last sumcheck-family call
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    static constexpr int kAux = 6;    // extra streams for independent MSM window classes of one batch
    static constexpr int kParts = 4;  // staggered parts of one MSM class (zk_msm.hip)
    // where one MSM batch runs: lane 0 = the ctx stream (blocking calls), lanes 1..2 = the asynchronous jobs
    // (zk_msm_g1_batch_async), which alternate between two sets of streams so that the latency-bound tail of one job
    // overlaps with the sort and the accumulation of the next.  Streams / events of the async lanes are created on first use.
    struct MsmLane {
        hipStream_t main = nullptr;
        hipStream_t aux[kAux] = {};
        hipEvent_t ev_fork = nullptr, ev_join[kAux] = {}, ev_part[kParts] = {};
        std::vector<hipEvent_t> ev_cls;  // completion event per MSM window class of a batch
        std::vector<hipEvent_t> ev_sort;  // end of the sort phase per class: the classes of a batch share ONE level-1 scratch region (zk_msm.hip)
        bool ready = false;
        // async lanes: growable arenas and pinned staging of their own (kept between jobs, like the ctx's scratch), and
        // whether a job currently owns the lane
        Arena mem[10];
        void* pinned = nullptr;
        size_t pinned_cap = 0;
        bool busy = false;
    };
    static constexpr int kLanes = 4;  // lane 0 + up to three asynchronous jobs in flight
    MsmLane lanes[kLanes];
    hipEvent_t ev_async_in = nullptr;  // ctx stream -> job stream: the scalars of a job are produced on the ctx stream
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
    void* h_comm = nullptr;  // pinned staging of zk_d_msm (its own block: the MSM pass may re-allocate h_pinned)
    size_t h_comm_cap = 0;
};

namespace zk {

// Experiment / diagnostics knobs of the whole library in ONE place.  Defaults are the shipped configuration; the only
// ways to change them are zk_dbg_tune(key, value) (tests, tools) and the ZKHIP_TUNE="key=value,key=value" environment
// variable, read once by zk::tuning() (zk_api.cpp) -- no other getenv in the product code.
struct Tuning {
    // sumcheck family (zk_fr.hip)
    long sc_pass_wg = 0;      // workgroups per CU of the HBM passes (0: 2 product / 4 others)
    long sc_local_g = 256;    // workgroups of a local stage
    long sc_local_threads = 1024;  // threads of a local-stage workgroup (256: fits beside two accumulation workgroups of an MSM pass -- EXPERIMENT, profiles/r06q)
    long sc_ts = 0;           // 1: in-kernel stage timestamps of the local launches on stderr, 2: host-side phases, 3: HIP events (zk_sumcheck_last_timing)
    long sc_xcd = 1;          // XCD-aware slice map of the local stages
    long sc_kf = 4;           // rounds per flat fold pass (0: round-by-round passes)
    long sc_plain_flat = 1;   // plain sumcheck passes in flat form
    long sc_kp = 2;           // rounds per product pass
    long sc_k0 = 3;           // rounds per single-table pass
    long sc_flat_wg = 64;     // workgroups per CU, flat fold
    long sc_plain_wg = 3;     // workgroups per CU, flat plain pass
    long sc_pre = 1;          // first round of a full-size local stage straight out of the table
    long sc_pinned_out = 1;   // results written straight into pinned host memory
    long sc_handover = 1;     // the pass before a multi-workgroup local stage stores that stage's slices contiguously
    long sc_t1_device = 0;    // TEST SWITCH: t1 = sum f_hi g_hi of EVERY round computed on the device (never derived)
    // MSM (zk_msm.hip)
    long msm_table_dc = 0;    // window-table width delta (sweeps)
    long msm_qstep = 2;       // window-class quantisation step of batches
    long msm_tile = 0;        // sorted entries per lane of k_accum_tiles (0: auto)
    long msm_pair = 0;        // k_finish pairs the planes
    long msm_fixq = 65536;    // buckets per class up to which the fix-up runs on quads
    long msm_quad = 32768;    // additions per reduction pass up to which a quad of lanes shares one
    long msm_stage = -1;      // level-1 scatter through LDS (-1: by row length)
    long msm_split = 1;       // staggered parts of the largest class
    long msm_np = 0;          // sort partitions per row (0: auto)
    long msm_fused_min = 0;   // window-table rows of >= 2^k entries are partitioned straight from the scalars (0: k = 23; -1: never -- the round-5 path)
    long msm_share_l1 = 1;    // the classes of a batch share one level-1 sort scratch (their sort phases run in sequence; 0: a region per class)
    long msm_idx_ahead = 1;   // k_accum_tiles: the sorted index of the entry after next is fetched one iteration early (0: the round-5 loop)
    long msm_tab_spt = 0;     // fused level 1: 1 forces one scalar per thread (A/B)
    long msm_l2_tiled = 0;    // level 2 in LDS-staged tiles for partitions of >= this many entries (0: 8192, rows of >= 2^23 entries; -1: never)
    long msm_debug = 0;       // class geometry on stderr
    long msm_serial = 0;      // all classes on the ctx stream
    long msm_size_classes = 1;  // window-table items: one class per power-of-two length
    long srs_table_batched = 1;  // G1 window tables: normalise with Montgomery's trick (one inversion per 64 records; 0: one per record, the round-5 kernel)
    long srs_table_rec = 0;   // G1 window tables built from now on: bytes per record (0: by free memory, see srs_precompute | 96: packed | 128: one record per 128-B line -- faster gathers, +33 % table memory)
    long msm_size_class_min = 16;  // window-table items of up to 2^k points (and one table width) share one size class: fewer launch chains for the short items of a batch
    long msm_small_table_widths = 1;  // zk_srs_precompute's own pick: 12 bits up to 2^10 points, 14 bits for 2^11 .. 2^14 (0: log2 n + 2)
    long msm_share = 100;     // EXPERIMENT (profiles/r05g): percent of the resident workgroup slots k_accum_tiles may fill (persistent grid below 100)
    // SRS / PSS maps on points (zk_srs.hip)
    long g1_map_by_column = 1;  // zk_g1_apply_matrix: one lane per (output, column) when there are few outputs of many terms (0: always one lane per output)
};
Tuning& tuning();
int tune_set(const char* key, long value);  // 0, or ZK_ERR_INVALID for an unknown key

int fail(zk_ctx* ctx, int code, const char* fmt, ...);
int hip_fail(zk_ctx* ctx, hipError_t e, const char* what);
// returns device scratch of at least `bytes` (slot 0..7), or nullptr after recording the error
void* scratch(zk_ctx* ctx, int slot, size_t bytes);
void* pinned(zk_ctx* ctx, size_t bytes);
// hipMalloc that drops the ctx's parked zk_free blocks and retries once on out-of-memory
hipError_t device_alloc(zk_ctx* ctx, void** out, size_t bytes, bool pool_locked = false);
size_t pool_trim(zk_ctx* ctx);
int msm_lanes_reserve(zk_ctx* ctx, int lane, const uint64_t* caps10, uint64_t pinned_cap);  // zk_msm.hip: grow an asynchronous lane's arenas

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
int multilinear_batch(zk_ctx* ctx, const zk_sc_item* items, size_t count);
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
// asynchronous form: enqueue on one of the ctx's job lanes and return; msm_job_wait runs the host chains, writes the
// results (18 u64 per item) and releases the job.  The scalars must stay valid until then.
int msm_g1_batch_async(zk_ctx* ctx, const MsmItem* items, size_t count, zk_msm_job** job);
int msm_job_wait(zk_ctx* ctx, zk_msm_job* job, uint64_t* h_out);
void msm_lanes_destroy(zk_ctx* ctx);
int msm_g2_batch(zk_ctx* ctx, const MsmItem* items, size_t count, uint64_t* h_out);  // h_out: 36 u64 per item
int srs_pack_g2(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out);
int msm_g1(zk_ctx* ctx, const zk_srs* srs, size_t offset, const void* d_scalars, size_t n, uint64_t* h_out);
int srs_pack(zk_ctx* ctx, const void* h_bases, size_t stride, size_t n, zk_srs** out);
int srs_precompute(zk_ctx* ctx, zk_srs* srs, int c, int record_bytes);  // record_bytes 0: the default (tuning knob srs_table_rec)
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
