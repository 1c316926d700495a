// zk_comm.cpp -- the party exchanges of the path behind the C ABI: an RCCL communicator inside the ctx.
//
// The reference moves every message through a TCP star (mpc-net) wrapped by the typed adapter
// dist-primitive/src/utils/serializing_net.rs:11-141 -- five patterns: gather to the leader (:11-39),
// gather to any receiver (:41-72), scatter from the leader (:74-96), scatter from any sender (:98-122)
// and leader_compute = gather + public map + scatter (:128-141).  On one node the party axis is the GPU
// axis, so they become collectives on raw Montgomery limbs in HBM over xGMI (no serialisation, no
// compression): zk_gather / zk_scatter for the four plain patterns, zk_allgather (+ the public map
// replicated on every party) for leader_compute, zk_alltoall for the looped dynamic scatters of
// dacc_product.rs:94-104,155-203.  All of them are enqueued on the ctx stream.
//
// RCCL is resolved at run time (dlopen of librccl.so.1, re-using the copy a host framework already
// mapped): libzkhip.so has no link-time dependency on it and loads on machines without it.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "zk_ctx.hpp"

// The handful of RCCL declarations this file needs, stated locally (ABI of nccl.h 2.x as shipped by RCCL): building
// libzkhip.so does not require the RCCL headers, loading it does not require the library.
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct {
    char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;  // (an int, not a one-value enum: RCCL returns codes 1..8 through it)
enum { ncclSuccess = 0 };  // every other value is an error; the text comes from ncclGetErrorString
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
}

namespace zk {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional (absent from very old builds): falls back to CommDestroy
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

static Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // a copy that is already mapped (torch ships its own) must be the one we use: one RCCL, one HIP runtime
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            const char* e = dlerror();  // (a second call would return NULL: the first one clears the error)
            r.err = std::string("librccl.so.1 not found: ") + (e ? e : "");
            return;
        }
        bool ok = true;
        auto sym = [&](const char* n) {
            void* p = dlsym(r.handle, n);
            if (!p) {
                ok = false;
                r.err = std::string("RCCL symbol missing: ") + n;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) r.handle = nullptr;
    });
    return r.handle ? &r : nullptr;
}

static int rccl_missing(zk_ctx* ctx) { return fail(ctx, ZK_ERR_COMM, "RCCL unavailable (librccl.so.1 could not be loaded)"); }

#define ZK_NCCL(ctx, R, call)                                                                 \
    do {                                                                                       \
        ncclResult_t _r = (call);                                                              \
        if (_r != ncclSuccess) return zk::fail(ctx, ZK_ERR_COMM, "%s: %s", #call, (R)->GetErrorString(_r)); \
    } while (0)

static int need_comm(zk_ctx* ctx, Rccl** r) {
    if (!ctx) return ZK_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, ZK_ERR_COMM, "no communicator: call zk_comm_init first");
    *r = rccl();
    if (!*r) return rccl_missing(ctx);
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    return ZK_OK;
}

// zk_d_msm's staging -- device scratch slot 7 and the pinned block h_comm -- for batches of up to D_MSM_PREALLOC_ITEMS results is
// taken when the communicator is created, so the call itself cannot run out of memory before it has joined the exchange (a
// party that returns early leaves its peers waiting in the collective).  Larger batches grow the blocks at call time.
constexpr size_t D_MSM_PREALLOC_ITEMS = 256;
static size_t d_msm_stage_bytes(size_t count, int world) { return (count * 144 + 16) * (size_t)(world + 1); }
static int d_msm_reserve(zk_ctx* ctx, size_t stage) {
    if (!scratch(ctx, 7, stage)) return ZK_ERR_OOM;
    if (ctx->h_comm_cap < stage) {
        if (ctx->h_comm) hipHostFree(ctx->h_comm);
        ctx->h_comm = nullptr, ctx->h_comm_cap = 0;
        const size_t cap = std::max<size_t>(stage, 4096);
        if (hipHostMalloc(&ctx->h_comm, cap, hipHostMallocDefault) != hipSuccess) {
            ctx->h_comm = nullptr;
            return ZK_ERR_OOM;
        }
        ctx->h_comm_cap = cap;
    }
    return ZK_OK;
}
// An error while ENQUEUEING an exchange (the HIP stream or the communicator is unusable): this party cannot take part any
// more.  Abort its communicator so that its resources are released and the peers' RCCL sees a dead rank (their collectives
// end with an error where the transport detects it, and with the job where it does not -- the reference's `unwrap()` panic
// does the same); the ctx is left without a communicator, every later exchange returns ZK_ERR_COMM at once.
static void comm_abort(zk_ctx* ctx, Rccl* r) {
    if (!ctx->comm) return;
    if (r->CommAbort) r->CommAbort((ncclComm_t)ctx->comm);
    else r->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_comm_unique_id(uint8_t h_id[ZK_COMM_ID_BYTES]) {
    static_assert(ZK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!h_id) return ZK_ERR_INVALID;
    Rccl* r = rccl();
    if (!r) return ZK_ERR_COMM;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return ZK_ERR_COMM;
    std::memcpy(h_id, id.internal, ZK_COMM_ID_BYTES);
    return ZK_OK;
}

int zk_comm_init(zk_ctx* ctx, int rank, int world, const uint8_t h_id[ZK_COMM_ID_BYTES]) {
    if (!ctx) return ZK_ERR_INVALID;
    if (!h_id || world < 1 || rank < 0 || rank >= world) return fail(ctx, ZK_ERR_INVALID, "zk_comm_init: bad rank / world / id");
    if (ctx->comm) return fail(ctx, ZK_ERR_INVALID, "zk_comm_init: the ctx already has a communicator");
    Rccl* r = rccl();
    if (!r) return rccl_missing(ctx);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(id.internal, h_id, ZK_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    ZK_NCCL(ctx, r, r->CommInitRank(&c, world, id, rank));
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    if (d_msm_reserve(ctx, d_msm_stage_bytes(D_MSM_PREALLOC_ITEMS, world))) {
        comm_abort(ctx, r);
        return fail(ctx, ZK_ERR_OOM, "zk_comm_init: no staging memory for zk_d_msm");
    }
    return ZK_OK;
}

int zk_comm_init_all(zk_ctx* const* ctxs, int world) {
    if (!ctxs || world < 1) return ZK_ERR_INVALID;
    for (int i = 0; i < world; i++)
        if (!ctxs[i] || ctxs[i]->comm) return ZK_ERR_INVALID;
    Rccl* r = rccl();
    if (!r) return rccl_missing(ctxs[0]);
    std::vector<int> devs(world);
    for (int i = 0; i < world; i++) devs[i] = ctxs[i]->device;
    std::vector<ncclComm_t> comms(world, nullptr);
    ZK_NCCL(ctxs[0], r, r->CommInitAll(comms.data(), world, devs.data()));
    for (int i = 0; i < world; i++) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_world = world;
    }
    int rc = ZK_OK;
    for (int i = 0; i < world && !rc; i++) {
        hipSetDevice(ctxs[i]->device);
        if (d_msm_reserve(ctxs[i], d_msm_stage_bytes(D_MSM_PREALLOC_ITEMS, world))) rc = fail(ctxs[i], ZK_ERR_OOM, "zk_comm_init_all: no staging memory for zk_d_msm");
    }
    if (rc)
        for (int i = 0; i < world; i++) comm_abort(ctxs[i], r);
    return rc;
}

int zk_comm_abort(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_INVALID;
    if (!ctx->comm) return ZK_OK;
    Rccl* r = rccl();
    if (r) comm_abort(ctx, r);
    return ZK_OK;
}

int zk_comm_destroy(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_INVALID;
    if (!ctx->comm) return ZK_OK;
    Rccl* r = rccl();
    if (r) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
        r->CommDestroy((ncclComm_t)ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    return ZK_OK;
}

int zk_comm_rank(const zk_ctx* ctx) { return ctx ? ctx->comm_rank : -1; }
int zk_comm_size(const zk_ctx* ctx) { return ctx ? ctx->comm_world : -1; }

int zk_allgather(zk_ctx* ctx, const void* d_send, size_t bytes, void* d_recv) {
    Rccl* r = nullptr;
    int rc = need_comm(ctx, &r);
    if (rc) return rc;
    if (bytes == 0) return ZK_OK;
    if (!d_send || !d_recv) return fail(ctx, ZK_ERR_INVALID, "null argument");
    ZK_NCCL(ctx, r, r->AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));
    return ZK_OK;
}

int zk_alltoall(zk_ctx* ctx, const void* d_send, size_t bytes_per_peer, void* d_recv) {
    Rccl* r = nullptr;
    int rc = need_comm(ctx, &r);
    if (rc) return rc;
    if (bytes_per_peer == 0) return ZK_OK;
    if (!d_send || !d_recv) return fail(ctx, ZK_ERR_INVALID, "null argument");
    const int w = ctx->comm_world;
    ZK_NCCL(ctx, r, r->GroupStart());
    for (int p = 0; p < w; p++) {
        ncclResult_t a = r->Send((const char*)d_send + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, p, (ncclComm_t)ctx->comm, ctx->stream);
        ncclResult_t b = r->Recv((char*)d_recv + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, p, (ncclComm_t)ctx->comm, ctx->stream);
        if (a != ncclSuccess || b != ncclSuccess) {
            r->GroupEnd();
            return fail(ctx, ZK_ERR_COMM, "ncclSend/ncclRecv: %s", r->GetErrorString(a != ncclSuccess ? a : b));
        }
    }
    ZK_NCCL(ctx, r, r->GroupEnd());
    return ZK_OK;
}

int zk_gather(zk_ctx* ctx, const void* d_send, size_t bytes, int root, void* d_recv) {
    Rccl* r = nullptr;
    int rc = need_comm(ctx, &r);
    if (rc) return rc;
    const int w = ctx->comm_world, me = ctx->comm_rank;
    if (root < 0 || root >= w) return fail(ctx, ZK_ERR_INVALID, "zk_gather: bad root");
    if (bytes == 0) return ZK_OK;
    if (!d_send || (me == root && !d_recv)) return fail(ctx, ZK_ERR_INVALID, "null argument");
    ZK_NCCL(ctx, r, r->GroupStart());
    ncclResult_t e = r->Send(d_send, bytes, ncclUint8, root, (ncclComm_t)ctx->comm, ctx->stream);
    if (me == root)
        for (int p = 0; p < w && e == ncclSuccess; p++) e = r->Recv((char*)d_recv + (size_t)p * bytes, bytes, ncclUint8, p, (ncclComm_t)ctx->comm, ctx->stream);
    ncclResult_t g = r->GroupEnd();
    if (e != ncclSuccess || g != ncclSuccess) return fail(ctx, ZK_ERR_COMM, "zk_gather: %s", r->GetErrorString(e != ncclSuccess ? e : g));
    return ZK_OK;
}

int zk_scatter(zk_ctx* ctx, const void* d_send, size_t bytes, int root, void* d_recv) {
    Rccl* r = nullptr;
    int rc = need_comm(ctx, &r);
    if (rc) return rc;
    const int w = ctx->comm_world, me = ctx->comm_rank;
    if (root < 0 || root >= w) return fail(ctx, ZK_ERR_INVALID, "zk_scatter: bad root");
    if (bytes == 0) return ZK_OK;
    if (!d_recv || (me == root && !d_send)) return fail(ctx, ZK_ERR_INVALID, "null argument");
    ZK_NCCL(ctx, r, r->GroupStart());
    ncclResult_t e = r->Recv(d_recv, bytes, ncclUint8, root, (ncclComm_t)ctx->comm, ctx->stream);
    if (me == root)
        for (int p = 0; p < w && e == ncclSuccess; p++) e = r->Send((const char*)d_send + (size_t)p * bytes, bytes, ncclUint8, p, (ncclComm_t)ctx->comm, ctx->stream);
    ncclResult_t g = r->GroupEnd();
    if (e != ncclSuccess || g != ncclSuccess) return fail(ctx, ZK_ERR_COMM, "zk_scatter: %s", r->GetErrorString(e != ncclSuccess ? e : g));
    return ZK_OK;
}

// d_msm end to end (dist-primitive/src/dmsm.rs:9-43): the local MSMs of the batch (:19-24), the gather of
// the 144-byte results (:29) as one all-gather over xGMI, and the leader's public map unpack2 -> sum ->
// pack_from_public (:30-39) replicated on every party for its own slot:  out_k = sum_i coeff_i * C_{i,k}.
// h_lambda (optional, Montgomery): this party's scalars are multiplied by it on the device first
// (MSM(b, lambda s) = lambda MSM(b, s)); with lambda_p = sum_j unpack2[j][p] every coeff_i collapses to the
// one pack coefficient c_p and the map is 7 point additions and a single scalar multiplication.
//
// A party whose LOCAL part fails (length mismatch, out of memory, a HIP error) still joins the exchange: every
// payload carries a status word, so all parties return an error instead of the healthy ones blocking forever in
// the collective (the reference's `unwrap()` panic takes the whole job down; a silent distributed hang would not).
// The staging memory of the exchange (device slot 7 + a pinned block of the ctx's own) exists since zk_comm_init for batches
// of up to 256 results (larger ones grow it BEFORE the local MSMs), so a party that runs out of memory in its MSMs can still
// publish its status.  Paths on which a party does NOT join: (1) no communicator, or an empty batch, on this party only (a
// caller error: nobody else is told); (2) a batch beyond 256 results whose staging cannot be grown; (3) an error of the HIP
// runtime or of RCCL while enqueueing the exchange itself.  In (2) and (3) the party returns an error AND aborts its
// communicator (ncclCommAbort), the library's form of the reference's `unwrap()` panic: its peers are not left with a
// half-alive rank, and every later exchange on this ctx fails at once with ZK_ERR_COMM.
// The MSM results reach the all-gather through pinned host memory: the last step of an MSM (the ~40-step bit-plane /
// window chain and the normalisation) runs on the host by design, so the 144-byte points exist on the
// host first; the payload is count x 144 + 16 bytes.
int zk_d_msm(zk_ctx* ctx, size_t count, const zk_srs* const* srs, const size_t* offsets, const void* const* d_scalars, const size_t* n,
             const uint64_t* h_lambda, const uint64_t* h_coeffs, uint64_t* h_out) {
    Rccl* r = nullptr;
    int rc = need_comm(ctx, &r);
    if (rc) return rc;  // (no communicator: nobody is waiting for this party)
    if (count == 0) return ZK_OK;
    const int w = ctx->comm_world;
    const size_t bytes = count * 144 + 16;  // results + status word (own 16-byte slot keeps the points aligned)
    // ---- staging of the exchange first: whatever happens below, the status can travel ----
    const size_t stage = bytes * (size_t)(w + 1);
    if (d_msm_reserve(ctx, stage)) {  // (only batches beyond D_MSM_PREALLOC_ITEMS can get here without memory)
        comm_abort(ctx, r);
        return fail(ctx, ZK_ERR_OOM, "zk_d_msm: no staging memory for the exchange (%zu bytes): this party cannot join it; its communicator was aborted", stage);
    }
    char* d = (char*)scratch(ctx, 7, stage);
    char* hp = (char*)ctx->h_comm;
    // ---- local part; any failure is carried into the exchange as `local_rc` ----
    int local_rc = ZK_OK;
    std::vector<uint64_t> local(count * 18 + 2, 0);
    std::vector<void*> scaled;
    if (!srs || !d_scalars || !n || !h_coeffs || !h_out) local_rc = fail(ctx, ZK_ERR_INVALID, "null argument");
    for (size_t k = 0; k < count && !local_rc; k++) {  // lengths first, before any device work
        if (!srs[k]) {
            local_rc = fail(ctx, ZK_ERR_INVALID, "null srs");
            continue;
        }
        const size_t off = offsets ? offsets[k] : 0, avail = srs[k]->n - std::min(off, srs[k]->n);  // (no sum that could wrap)
        if (n[k] > avail)
            local_rc = fail(ctx, ZK_ERR_LENGTH, "d_msm item %zu: %zu scalars but only %zu bases from offset %zu", k, n[k], avail, off);
    }
    if (!local_rc) {
        std::vector<MsmItem> items(count);
        const uint64_t zero[4] = {0, 0, 0, 0};
        for (size_t k = 0; k < count && !local_rc; k++) {
            const void* sc = d_scalars[k];
            if (h_lambda && n[k]) {
                void* t = nullptr;
                local_rc = zk_malloc(ctx, n[k] * 32, &t);
                if (!local_rc) {
                    scaled.push_back(t);
                    local_rc = fr_axpb(ctx, nullptr, sc, h_lambda, zero, t, n[k]);
                }
                sc = t;
            }
            items[k] = MsmItem{srs[k], offsets ? offsets[k] : 0, sc, n[k]};
        }
        if (!local_rc) local_rc = msm_g1_batch(ctx, items.data(), count, local.data());
        for (void* p : scaled) zk_free(ctx, p);
    }
    const std::string local_err = local_rc ? ctx->err : std::string();
    if (local_rc) std::fill(local.begin(), local.end(), 0);
    local[count * 18] = (uint64_t)(int64_t)local_rc;
    // ---- exchange (every party, healthy or not) ----
    std::memcpy(hp, local.data(), bytes);
    {
        hipError_t e = hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, ctx->stream);
        ncclResult_t g = e == hipSuccess ? r->AllGather(d, d + bytes, bytes, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream) : ncclSuccess;
        if (e == hipSuccess && g == ncclSuccess) e = hipMemcpyAsync(hp + bytes, d + bytes, bytes * w, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && g == ncclSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess || g != ncclSuccess) {  // this party is out of the exchange: do not leave a half-dead rank behind
            const std::string why = g != ncclSuccess ? std::string("ncclAllGather: ") + r->GetErrorString(g) : std::string(hipGetErrorString(e));
            comm_abort(ctx, r);
            return fail(ctx, ZK_ERR_COMM, "zk_d_msm: the exchange failed on this party (%s); its communicator was aborted", why.c_str());
        }
    }
    const char* all = hp + bytes;
    if (local_rc) return fail(ctx, local_rc, "%s", local_err.c_str());
    for (int p = 0; p < w; p++) {
        int64_t st;
        std::memcpy(&st, all + (size_t)p * bytes + count * 144, 8);
        if (st) return fail(ctx, ZK_ERR_COMM, "zk_d_msm: party %d failed in its local MSMs (code %d)", p, (int)st);
    }
    // rows of the combination: item k <- the w points C_{0,k} .. C_{w-1,k}
    std::vector<uint64_t> rows((size_t)count * w * 18);
    for (size_t k = 0; k < count; k++)
        for (int p = 0; p < w; p++) std::memcpy(&rows[(k * w + p) * 18], all + (size_t)p * bytes + k * 144, 144);
    return g1_lincomb_batch_host(ctx, rows.data(), h_coeffs, (size_t)w, count, h_out);
}

}  // extern "C"
