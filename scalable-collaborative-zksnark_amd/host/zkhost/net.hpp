// Party-axis exchange layer of the C++ host mirror -- what replaces mpc-net's TCP star (mpc-net/src/lib.rs:35-287) and the
// typed adapter dist-primitive/src/utils/serializing_net.rs.
//
// Every exchange of the hot path is a star through party 0 whose leader function is a PUBLIC linear map, so
// on one node it is ONE all-gather of raw Montgomery limbs followed by the same map computed by every party for its own slot.
//
//   RcclNet        the communicator inside the ctx (zk_comm_init / zk_allgather / zk_alltoall): buffers stay in HBM, xGMI
//   LocalTestNet   all parties as threads of one process (LocalTestNet::simulate_network_round, mpc-net/src/multi.rs:268-362)
//   LeaderEchoNet  the no-`comm` fake (serializing_net.rs:144-264): the leader sees n copies of its own message
#pragma once
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "device.hpp"

namespace zkhost {

using Bytes = std::vector<uint8_t>;
enum class Echo { Slot0, Identity };  // what the leader-echo fake returns for a looped dynamic exchange (see LeaderEchoNet)

class Net {
  public:
    size_t n_parties = 1, party_id = 0;
    bool echo = false;                   // the no-`comm` fake: d_msm applies the public map where the reference applies it
    uint64_t upload = 0, download = 0;  // byte accounting like MPCNet::get_comm (multi.rs:378-387)
    virtual ~Net() {}
    bool is_leader() const { return party_id == 0; }

    // every party contributes `bytes` (same size everywhere); -> the contributions ordered by party id
    virtual std::vector<Bytes> all_gather(const void *data, size_t bytes) = 0;
    // party p sends chunks[q] to party q; -> what party q sent to me, q < n.  Replaces the reference's loops of dynamic
    // gathers / scatters over every root (dacc_product.rs:94-104,155-203).  Default: all-gather of the chunks, then select.
    virtual std::vector<Bytes> all_to_all(const std::vector<Bytes> &chunks, Echo = Echo::Slot0) {
        size_t per = chunks.empty() ? 0 : chunks[0].size();
        Bytes flat;
        for (auto &c : chunks) flat.insert(flat.end(), c.begin(), c.end());
        std::vector<Bytes> all = all_gather(flat.data(), flat.size()), out;
        for (size_t q = 0; q < n_parties; ++q) out.emplace_back(all[q].begin() + party_id * per, all[q].begin() + (party_id + 1) * per);
        return out;
    }
    void sync() {
        uint64_t z = 0;
        all_gather(&z, 8);
    }

    // device-buffer forms.  These defaults stage through the host methods above (thread / echo nets of the tests); RcclNet
    // overrides them with collectives that never leave HBM.
    virtual DevPtr all_gather_device(Ctx &be, const DevPtr &send, size_t bytes) {
        Bytes mine(bytes);
        be.download(mine.data(), send, bytes);
        std::vector<Bytes> parts = all_gather(mine.data(), bytes);
        DevPtr out = be.alloc(bytes * n_parties);
        for (size_t q = 0; q < n_parties; ++q) be.upload(out.at(q * bytes), parts[q].data(), bytes);
        return out;
    }
    virtual DevPtr all_to_all_device(Ctx &be, const DevPtr &send, size_t bytes_per_peer, Echo e = Echo::Slot0) {
        Bytes mine(bytes_per_peer * n_parties);
        be.download(mine.data(), send, mine.size());
        std::vector<Bytes> chunks;
        for (size_t q = 0; q < n_parties; ++q) chunks.emplace_back(mine.begin() + q * bytes_per_peer, mine.begin() + (q + 1) * bytes_per_peer);
        std::vector<Bytes> got = all_to_all(chunks, e);
        DevPtr out = be.alloc(bytes_per_peer * n_parties);
        for (size_t q = 0; q < n_parties; ++q) be.upload(out.at(q * bytes_per_peer), got[q].data(), bytes_per_peer);
        return out;
    }

    // typed helpers: raw limbs, no serialisation
    std::vector<FrVec> all_gather_fr(const FrVec &v) {
        std::vector<Bytes> got = all_gather(v.data(), 32 * v.size());
        std::vector<FrVec> out(n_parties, FrVec(v.size()));
        for (size_t q = 0; q < n_parties; ++q) std::memcpy(out[q].data(), got[q].data(), 32 * v.size());
        return out;
    }
    std::vector<G1Vec> all_gather_g1(const G1Vec &v) {
        std::vector<Bytes> got = all_gather(v.data(), 144 * v.size());
        std::vector<G1Vec> out(n_parties, G1Vec(v.size()));
        for (size_t q = 0; q < n_parties; ++q) std::memcpy(out[q].data(), got[q].data(), 144 * v.size());
        return out;
    }
    std::vector<FrVec> all_to_all_fr(const std::vector<FrVec> &chunks, Echo e = Echo::Slot0) {
        std::vector<Bytes> raw;
        for (auto &c : chunks) raw.emplace_back((const uint8_t *)c.data(), (const uint8_t *)c.data() + 32 * c.size());
        std::vector<Bytes> got = all_to_all(raw, e);
        std::vector<FrVec> out;
        for (auto &g : got) {
            FrVec v(g.size() / 32);
            std::memcpy(v.data(), g.data(), g.size());
            out.push_back(std::move(v));
        }
        return out;
    }

  protected:
    void count(size_t bytes) {
        upload += bytes * (n_parties - 1);
        download += bytes * (n_parties - 1);
    }
};

// serializing_net.rs:144-264: a gather returns n copies of the own message (:159-162 -> Echo::Slot0); for the looped dynamic
// scatters the callers substitute their local data per destination (dacc_product.rs:194-202 -> Echo::Identity).  Only
// party 0 is meaningful.
class LeaderEchoNet : public Net {
  public:
    explicit LeaderEchoNet(size_t n = 8) {
        n_parties = n, party_id = 0, echo = true;
    }
    std::vector<Bytes> all_gather(const void *data, size_t bytes) override {
        count(bytes);
        return std::vector<Bytes>(n_parties, Bytes((const uint8_t *)data, (const uint8_t *)data + bytes));
    }
    std::vector<Bytes> all_to_all(const std::vector<Bytes> &chunks, Echo e = Echo::Slot0) override {
        size_t tot = 0;
        for (auto &c : chunks) tot += c.size();
        count(tot / std::max<size_t>(chunks.size(), 1));
        if (e == Echo::Identity) return chunks;
        return std::vector<Bytes>(n_parties, chunks[0]);
    }
    // the fabricated copies are made in HBM (the leader's data never visits the host)
    DevPtr all_gather_device(Ctx &be, const DevPtr &send, size_t bytes) override {
        count(bytes);
        DevPtr out = be.alloc(bytes * n_parties);
        for (size_t q = 0; q < n_parties; ++q) be.copy_d2d(out.at(q * bytes), send, bytes);
        return out;
    }
    DevPtr all_to_all_device(Ctx &be, const DevPtr &send, size_t bytes_per_peer, Echo e = Echo::Slot0) override {
        count(bytes_per_peer);
        DevPtr out = be.alloc(bytes_per_peer * n_parties);
        if (e == Echo::Identity) be.copy_d2d(out, send, bytes_per_peer * n_parties);
        else
            for (size_t q = 0; q < n_parties; ++q) be.copy_d2d(out.at(q * bytes_per_peer), send, bytes_per_peer);
        return out;
    }
};

// all parties as threads of one process
class LocalHub {
  public:
    explicit LocalHub(size_t n) : n_(n), slots_(n) {}
    size_t n() const { return n_; }
    // a reusable barrier; a party that failed calls abort() so that the others do not wait for ever
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        if (aborted_) throw std::runtime_error("LocalTestNet: another party failed");
        size_t gen = gen_;
        if (++arrived_ == n_) {
            arrived_ = 0, ++gen_;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return gen_ != gen || aborted_; });
            if (aborted_) throw std::runtime_error("LocalTestNet: another party failed");
        }
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m_);
        aborted_ = true;
        cv_.notify_all();
    }
    std::vector<std::vector<Bytes>> &slots() { return slots_; }

  private:
    size_t n_, arrived_ = 0, gen_ = 0;
    bool aborted_ = false;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::vector<Bytes>> slots_;
};

class LocalTestNet : public Net {
  public:
    LocalTestNet(LocalHub &hub, size_t party) : hub_(hub) {
        n_parties = hub.n(), party_id = party;
    }
    std::vector<Bytes> all_gather(const void *data, size_t bytes) override {
        hub_.slots()[party_id] = {Bytes((const uint8_t *)data, (const uint8_t *)data + bytes)};
        hub_.wait();
        std::vector<Bytes> out;
        for (size_t q = 0; q < n_parties; ++q) out.push_back(hub_.slots()[q][0]);
        hub_.wait();
        count(bytes);
        return out;
    }
    std::vector<Bytes> all_to_all(const std::vector<Bytes> &chunks, Echo = Echo::Slot0) override {
        hub_.slots()[party_id] = chunks;
        hub_.wait();
        std::vector<Bytes> out;
        for (size_t q = 0; q < n_parties; ++q) out.push_back(hub_.slots()[q][party_id]);
        hub_.wait();
        size_t tot = 0;
        for (auto &c : chunks) tot += c.size();
        count(tot / std::max<size_t>(chunks.size(), 1));
        return out;
    }
    // run fn(party, net) for every party on its own thread (mpc-net/src/multi.rs:330-352); the first failure is rethrown
    static void simulate_network_round(size_t n, const std::function<void(size_t, LocalTestNet &)> &fn) {
        LocalHub hub(n);
        std::vector<std::thread> th;
        std::mutex em;
        std::exception_ptr err;
        for (size_t p = 0; p < n; ++p)
            th.emplace_back([&, p] {
                try {
                    LocalTestNet net(hub, p);
                    fn(p, net);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(em);
                    if (!err) err = std::current_exception();
                    hub.abort();
                }
            });
        for (auto &t : th) t.join();
        if (err) std::rethrow_exception(err);
    }

  private:
    LocalHub &hub_;
};

// The C-ABI communicator: the exchanges run on the ctx stream over RCCL and the buffers stay in HBM
// (`all_gather_device` moves no byte over PCIe).  The host-typed methods stage through a device buffer.
class RcclNet : public Net {
  public:
    // the ctx already carries its communicator (zk_comm_init or zk_comm_init_all)
    explicit RcclNet(Ctx &ctx) : ctx_(ctx) {
        int r = zk_comm_rank(ctx.handle()), w = zk_comm_size(ctx.handle());
        if (r < 0 || w <= 0) throw ZkError(ZK_ERR_COMM, "RcclNet: the ctx has no communicator");
        party_id = r, n_parties = w;
    }
    // one process per GPU: rank 0 creates the id (zk_comm_unique_id) and hands it to the others out of band
    RcclNet(Ctx &ctx, int rank, int world, const uint8_t id[ZK_COMM_ID_BYTES]) : ctx_(ctx) {
        ctx.check(zk_comm_init(ctx.handle(), rank, world, id));
        party_id = rank, n_parties = world;
    }
    Ctx &ctx() { return ctx_; }
    DevPtr all_gather_device(Ctx &be, const DevPtr &send, size_t bytes) override {
        same(be);
        count(bytes);
        DevPtr out = ctx_.alloc(bytes * n_parties);
        ctx_.check(zk_allgather(ctx_.handle(), send.get(), bytes, out.get()));
        return out;
    }
    DevPtr all_to_all_device(Ctx &be, const DevPtr &send, size_t bytes_per_peer, Echo = Echo::Slot0) override {
        same(be);
        count(bytes_per_peer);
        DevPtr out = ctx_.alloc(bytes_per_peer * n_parties);
        ctx_.check(zk_alltoall(ctx_.handle(), send.get(), bytes_per_peer, out.get()));
        return out;
    }
    std::vector<Bytes> all_gather(const void *data, size_t bytes) override {
        if (!bytes) return std::vector<Bytes>(n_parties);
        DevPtr d = ctx_.alloc(bytes);
        ctx_.upload(d, data, bytes);
        DevPtr all = all_gather_device(ctx_, d, bytes);
        Bytes flat(bytes * n_parties);
        ctx_.download(flat.data(), all, flat.size());
        std::vector<Bytes> out;
        for (size_t q = 0; q < n_parties; ++q) out.emplace_back(flat.begin() + q * bytes, flat.begin() + (q + 1) * bytes);
        return out;
    }
    std::vector<Bytes> all_to_all(const std::vector<Bytes> &chunks, Echo = Echo::Slot0) override {
        size_t per = chunks.empty() ? 0 : chunks[0].size();
        if (!per) return std::vector<Bytes>(n_parties);
        Bytes flat;
        for (auto &c : chunks) flat.insert(flat.end(), c.begin(), c.end());
        DevPtr d = ctx_.alloc(flat.size());
        ctx_.upload(d, flat.data(), flat.size());
        DevPtr got = all_to_all_device(ctx_, d, per);
        ctx_.download(flat.data(), got, flat.size());
        std::vector<Bytes> out;
        for (size_t q = 0; q < n_parties; ++q) out.emplace_back(flat.begin() + q * per, flat.begin() + (q + 1) * per);
        return out;
    }
    // zk_d_msm applies when the exchange of d_msm can run inside the library
    bool owns(const Ctx &be) const { return &be == &ctx_; }
    // byte accounting of an exchange the library ran itself (zk_d_msm: one all-gather of `bytes` per party)
    void account(size_t bytes) { count(bytes); }

  private:
    void same(const Ctx &be) const {
        if (&be != &ctx_) throw ZkError(ZK_ERR_COMM, "RcclNet: the communicator lives in its own ctx");
    }
    Ctx &ctx_;
};

}  // namespace zkhost
