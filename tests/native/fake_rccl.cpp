// TEST DOUBLE of librccl.so.1 for ONE process whose ranks are host threads that may share a GPU.
//
// No multi-GPU node was available while this library was built, and the real RCCL refuses two ranks on one device: the party
// exchanges of csrc/zk_comm.cpp (zk_allgather, zk_alltoall, zk_gather, zk_scatter, zk_d_msm) and the nets above them
// (zkhost::RcclNet, zkhip.net.RcclNet) had only ever run at world size 1.  This file implements the handful of RCCL entry
// points the library resolves with dlopen (csrc/zk_comm.cpp `struct Rccl`) with the SEMANTICS of the real ones -- rank order,
// byte counts, in-order matching of grouped sends and receives, buffers in device memory, work ordered on the caller's
// stream -- by a rendezvous of the ranks' host threads and hipMemcpyAsync between their buffers.  Put its directory first in
// LD_LIBRARY_PATH and every zk_comm_* call of the process talks to it: the whole 8-party protocol then runs over "RCCL" on a
// one-GPU box, and what is tested is everything of OURS around the collectives (offsets, sizes, ordering, status words, error
// propagation, the hosts' nets).  What it cannot test is the wire.  It is test infrastructure: nothing in the product links it.
//
// Supported: ncclCommInitAll (all ranks in this process) and ncclCommInitRank (threads of ONE process sharing the id);
// uint8 payloads (all the library sends).  A rank that calls ncclCommAbort wakes every waiting rank with an error.
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
}

namespace {
enum { kOk = 0, kSystemError = 2, kInvalidArgument = 4, kInvalidUsage = 5 };

struct SendOp {
    int peer;
    const void* buf;
    size_t bytes;
};
struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long gen = 0;
    bool aborted = false;
    // what every rank published for the exchange in flight
    std::vector<const void*> send;
    std::vector<size_t> bytes;
    std::vector<std::vector<SendOp>> sends;
    explicit Group(int w) : world(w), send(w), bytes(w), sends(w) {}
    // reusable barrier; false = the communicator was aborted
    bool wait() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        unsigned long g = gen;
        if (++arrived == world) {
            arrived = 0, ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g || aborted; });
        }
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
};
}  // namespace

struct ncclComm {
    Group* g;
    int rank;
    int device;
};

namespace {
std::mutex g_reg_m;
std::map<std::string, Group*> g_by_id;  // ncclCommInitRank: threads of one process that share a unique id
unsigned long g_next_id = 1;

struct Pending {  // a rank's operations between ncclGroupStart and ncclGroupEnd (one caller thread = one rank)
    ncclComm* comm = nullptr;
    hipStream_t stream = nullptr;
    std::vector<SendOp> sends, recvs;
};
thread_local int t_depth = 0;
thread_local Pending t_pending;

ncclResult_t run_p2p(Pending& p) {
    if (!p.comm) return kOk;
    Group* g = p.comm->g;
    const int me = p.comm->rank;
    if (hipStreamSynchronize(p.stream) != hipSuccess) return kSystemError;  // my send buffers are final
    g->sends[me] = p.sends;
    if (!g->wait()) return kSystemError;
    std::vector<size_t> cursor(g->world, 0);  // in-order matching per (sender -> me), like the real library
    ncclResult_t rc = kOk;
    for (const SendOp& r : p.recvs) {
        const std::vector<SendOp>& theirs = g->sends[r.peer];
        size_t& c = cursor[r.peer];
        while (c < theirs.size() && theirs[c].peer != me) ++c;
        if (c >= theirs.size() || theirs[c].bytes != r.bytes) {
            rc = kInvalidUsage;  // an unmatched receive / a size mismatch hangs the real library: here it is reported
            break;
        }
        if (r.bytes && hipMemcpyAsync(const_cast<void*>(r.buf), theirs[c].buf, r.bytes, hipMemcpyDeviceToDevice, p.stream) != hipSuccess) rc = kSystemError;
        ++c;
    }
    if (hipStreamSynchronize(p.stream) != hipSuccess) rc = kSystemError;
    if (!g->wait()) return kSystemError;  // nobody reuses a send buffer before every receiver has copied it
    return rc;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return kInvalidArgument;
    std::lock_guard<std::mutex> lk(g_reg_m);
    std::memset(id->internal, 0, sizeof id->internal);
    std::snprintf(id->internal, sizeof id->internal, "zk-fake-rccl-%lu", g_next_id++);
    return kOk;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || rank < 0 || rank >= world) return kInvalidArgument;
    std::lock_guard<std::mutex> lk(g_reg_m);
    Group*& g = g_by_id[std::string(id.internal, strnlen(id.internal, sizeof id.internal))];
    if (!g) g = new Group(world);
    if (g->world != world) return kInvalidArgument;
    int dev = 0;
    (void)hipGetDevice(&dev);
    *comm = new ncclComm{g, rank, dev};
    return kOk;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int world, const int* devs) {
    if (!comms || world < 1) return kInvalidArgument;
    std::fprintf(stderr, "[fake_rccl] the test double of librccl.so.1 is in use: %d ranks as host threads of this process\n", world);
    Group* g = new Group(world);
    for (int i = 0; i < world; i++) comms[i] = new ncclComm{g, i, devs ? devs[i] : i};
    return kOk;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;  // (the Group is shared by the ranks and lives as long as the process: a test double)
    return kOk;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (comm) comm->g->abort();
    delete comm;
    return kOk;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !send || !recv) return kInvalidArgument;
    Group* g = comm->g;
    if (hipStreamSynchronize(stream) != hipSuccess) return kSystemError;  // my contribution is final
    g->send[comm->rank] = send;
    g->bytes[comm->rank] = count;
    if (!g->wait()) return kSystemError;
    ncclResult_t rc = kOk;
    for (int q = 0; q < g->world && rc == kOk; q++) {
        if (g->bytes[q] != count) rc = kInvalidUsage;  // (the real library would hang or corrupt: ranks must agree on the size)
        else if (count && hipMemcpyAsync((char*)recv + (size_t)q * count, g->send[q], count, hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = kSystemError;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) rc = kSystemError;
    if (!g->wait()) return kSystemError;
    return rc;
}

ncclResult_t ncclGroupStart() {
    if (t_depth++ == 0) t_pending = Pending();
    return kOk;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->g->world) return kInvalidArgument;
    const bool single = t_depth == 0;
    if (single) t_pending = Pending();
    if (t_pending.comm && t_pending.comm != comm) return kInvalidUsage;
    t_pending.comm = comm, t_pending.stream = stream;
    t_pending.sends.push_back({peer, buf, count});
    return single ? run_p2p(t_pending) : kOk;
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->g->world) return kInvalidArgument;
    const bool single = t_depth == 0;
    if (single) t_pending = Pending();
    if (t_pending.comm && t_pending.comm != comm) return kInvalidUsage;
    t_pending.comm = comm, t_pending.stream = stream;
    t_pending.recvs.push_back({peer, buf, count});
    return single ? run_p2p(t_pending) : kOk;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return kInvalidUsage;
    if (--t_depth > 0) return kOk;
    Pending p = t_pending;
    t_pending = Pending();
    return run_p2p(p);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case kOk: return "no error";
        case kSystemError: return "fake rccl: unhandled system error (HIP call failed or the communicator was aborted)";
        case kInvalidArgument: return "fake rccl: invalid argument";
        case kInvalidUsage: return "fake rccl: invalid usage (unmatched / mis-sized exchange)";
        default: return "fake rccl: unknown error";
    }
}

}  // extern "C"
