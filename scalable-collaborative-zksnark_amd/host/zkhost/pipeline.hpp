// The pipelined forms of the `dist-primitive` mirror: what a protocol driver uses to keep the GPU busy.
//
// Nothing on the reference's path consumes an MSM result on the device (challenges are pre-sampled, hyperplonk/src/
// dhyperplonk.rs:159-186), so every commit / open of a protocol step can QUEUE its MSMs, run them in ONE pipeline pass
// (zk_msm_g1_batch, or zk_msm_g1_batch_async beside the next step's kernels) and finish -- exchange + public map -- later.
// The `*_q` functions below run their kernels now, queue their MSMs and return a closure that produces the reference's
// return value once the queue has run; every party calls them, the queue and the closures in the same order.  Every output
// is bit-identical to the one-call-at-a-time forms of dist_primitive.hpp (same field / group elements).
// Lifetimes: a closure OWNS what it reads of the queue -- a shared handle on the pass its items were added to (MsmPass), so it
// stays valid after the MsmQueue object is gone or has been re-armed by a later pass, and it throws (instead of reading someone
// else's points) when called before that pass was collected.  Sharing-parameter values are captured by value.  The backend
// and the net are the party's long-lived objects (one Ctx and one Net per party for the whole run) and are referred to.
#pragma once
#include <functional>
#include <map>
#include <tuple>

#include "dist_primitive.hpp"

namespace zkhost {

// the results of ONE pass of a queue, shared between the queue and the closures whose items ran in it
struct MsmPass {
    G1Vec res;  // in insertion order
    bool collected = false;
    const G1 &at(size_t i) const {
        if (!collected) throw ZkError(ZK_ERR_INVALID, "MsmQueue: a result was read before its pass was run / finished");
        return res.at(i);
    }
};
using MsmPassRef = std::shared_ptr<MsmPass>;

class MsmQueue {
  public:
    std::vector<DevPtr> keep;  // buffers that must outlive the pass

    explicit MsmQueue(Ctx &be, bool dedup = true) : be_(be), dedup_(dedup), pass_(std::make_shared<MsmPass>()) {}
    // the pass the items added NOW will run in: what a closure keeps (by value) to read its results later
    MsmPassRef ticket() const { return pass_; }

    // -> indices of the items' results in `res`.  Identical items -- same level, same scalar buffer, same length -- are
    // computed ONCE (the two opens of V in step 2.d, dhyperplonk.rs:307-320, commit the same first quotient q_0 = V_hi - V_lo:
    // it does not depend on the opening point).  The queue holds the DevPtr of every item, so an address cannot be freed and
    // name another table while the queue collects.
    std::vector<size_t> add(const std::vector<SrsPtr> &srs, const std::vector<DevPtr> &bufs, const std::vector<size_t> &lens) {
        std::vector<size_t> idx;
        for (size_t i = 0; i < lens.size(); ++i) {
            auto key = std::make_tuple((const void *)srs[i]->handle(), (const void *)bufs[i].get(), lens[i]);
            auto it = dedup_ ? index_.find(key) : index_.end();
            size_t j;
            if (it == index_.end()) {
                j = lens_.size();
                srs_.push_back(srs[i]), bufs_.push_back(bufs[i]), lens_.push_back(lens[i]);
                if (dedup_) index_[key] = j;
            } else {
                j = it->second;
            }
            idx.push_back(j);
        }
        return idx;
    }
    // lambda buf for the pre-scaled d_msm, once per distinct scalar buffer
    DevPtr scale(const DevPtr &buf, const Fr &lam, size_t n) {
        auto key = std::make_tuple((const void *)buf.get(), n, lam.v[0], lam.v[1], lam.v[2], lam.v[3]);
        auto it = dedup_ ? scaled_.find(key) : scaled_.end();
        if (it != scaled_.end()) return it->second;
        DevPtr out = be_.fr_scale(buf, lam, n);
        if (dedup_) scaled_[key] = out;
        keep.push_back(buf);
        return out;
    }
    bool empty() const { return lens_.empty(); }
    void run() {
        close(lens_.empty() ? G1Vec{} : be_.msm_g1_batch(detail::raw(srs_), bufs_, lens_));
    }
    // the same in two halves: start() enqueues the pass and returns, finish() collects the points.  Between the two the
    // caller enqueues the NEXT step's kernels and does its own host work (the exchange closures of a step are host
    // arithmetic on a few hundred points, during which the GPU would otherwise idle).
    void start() {
        if (lens_.empty()) {
            close(G1Vec{});
        } else {
            job_ = be_.msm_g1_batch_async(detail::raw(srs_), bufs_, lens_);
        }
    }
    void finish() {
        if (job_.pending()) close(be_.msm_wait(job_));
    }

  private:
    // the pass is over: publish its points to the closures that hold its ticket, drop the owners and, with them, the keys
    // that named their addresses, and arm a fresh pass for whatever is added next
    void close(G1Vec &&points) {
        pass_->res = std::move(points);
        pass_->collected = true;
        pass_ = std::make_shared<MsmPass>();
        keep.clear(), index_.clear(), scaled_.clear(), srs_.clear(), bufs_.clear(), lens_.clear();
    }
    Ctx &be_;
    bool dedup_;
    std::vector<SrsPtr> srs_;
    std::vector<DevPtr> bufs_;
    std::vector<size_t> lens_;
    std::map<std::tuple<const void *, const void *, size_t>, size_t> index_;
    std::map<std::tuple<const void *, size_t, uint64_t, uint64_t, uint64_t, uint64_t>, DevPtr> scaled_;
    Ctx::MsmJob job_;
    MsmPassRef pass_;
};

namespace detail {
inline G1Vec pick(const MsmPassRef &pass, const std::vector<size_t> &idx) {
    G1Vec out;
    for (size_t i : idx) out.push_back(pass->at(i));
    return out;
}
}  // namespace detail

// d_msm (dmsm.rs:9-43) with its local MSMs queued
inline std::function<G1Vec()> d_msm_q(Ctx &be, MsmQueue &q, const std::vector<SrsPtr> &bases, const std::vector<DevPtr> &scalars, const std::vector<size_t> &lens,
                                      const PackedSharingParams &pp, Net &net, bool prescale = true) {
    if (bases.size() != scalars.size() || bases.size() != lens.size()) throw ZkError(ZK_ERR_INVALID, "d_msm: bases / scalars batch sizes differ (dmsm.rs:16)");
    size_t k = lens.size(), p = net.party_id, n = net.n_parties;
    if (!k) return [] { return G1Vec{}; };
    if (!prescale || net.echo) {
        std::vector<size_t> sl = q.add(bases, scalars, lens);
        FrVec coeffs = detail::canonical(pp.dmsm_coeffs(p));
        return [&be, pass = q.ticket(), &net, coeffs, sl, k] {
            std::vector<G1Vec> got = net.all_gather_g1(detail::pick(pass, sl));
            return be.g1_lincomb_batch(detail::by_item(got), coeffs, k);
        };
    }
    Fr lam = pp.lambda(p), cp = pp.c(p);
    std::vector<DevPtr> scaled;
    for (size_t i = 0; i < k; ++i) scaled.push_back(q.scale(scalars[i], lam, lens[i]));
    std::vector<size_t> sl = q.add(bases, scaled, lens);
    return [&be, pass = q.ticket(), &net, sl, k, n, cp] {
        std::vector<G1Vec> got = net.all_gather_g1(detail::pick(pass, sl));
        G1Vec sums = be.g1_lincomb_batch(detail::by_item(got), FrVec(n, Fr{{1, 0, 0, 0}}), k);
        return be.g1_lincomb_batch(sums, FrVec{cp.to_canonical()}, k);
    };
}

inline std::function<G1()> commit_q(MsmQueue &q, const PowersOfG &pg, const DevPtr &peval, size_t len) {
    std::vector<size_t> sl = q.add({detail::level_for(pg, len)}, {peval}, {len});
    return [pass = q.ticket(), sl] { return pass->at(sl[0]); };
}

inline std::function<G1Vec()> c_commit_q(Ctx &be, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals, const std::vector<size_t> &lens,
                                         const PackedSharingParams &pp, Net &net) {
    std::vector<SrsPtr> bases;
    for (size_t n : lens) bases.push_back(detail::level_for(pg, n * pp.l));  // dpoly_comm.rs:256-257
    return d_msm_q(be, q, bases, pevals, lens, pp, net);
}

// several d_commit (dpoly_comm.rs:276-297) in one pass and one exchange
inline std::function<G1Vec()> d_commit_many_q(Ctx &be, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals, const std::vector<size_t> &lens, Net &net) {
    size_t k = lens.size();
    if (!k) return [] { return G1Vec{}; };
    std::vector<SrsPtr> srs;
    for (size_t n : lens) srs.push_back(detail::level_for(pg, n));
    std::vector<size_t> sl = q.add(srs, pevals, lens);
    return [&be, pass = q.ticket(), &net, sl, k] {
        std::vector<G1Vec> got = net.all_gather_g1(detail::pick(pass, sl));
        return be.g1_lincomb_batch(detail::by_item(got), FrVec(net.n_parties, Fr{{1, 0, 0, 0}}), k);
    };
}

// several independent opens (dpoly_comm.rs:299-325 each): the fold rounds run now as one batched call, the commitments of
// all q_i are queued.  Opens of the SAME table share their first quotient q_0 = hi - lo (it does not depend on the point):
// with the queue's duplicate detection its commitment is one MSM for all of them.
struct OpensInFlight {
    std::vector<Fr> values;  // known as soon as the fold rounds ran (before the MSMs)
    std::function<std::vector<Opening>()> finish;
};
inline OpensInFlight open_many_q(Ctx &be, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals, const std::vector<size_t> &lens,
                                 const std::vector<FrVec> &points, size_t l = 1) {
    std::vector<ScRequest> reqs;
    for (size_t i = 0; i < lens.size(); ++i) {
        size_t n = Ctx::log2_exact(lens[i]);
        if (points[i].size() < n) throw ZkError(ZK_ERR_INVALID, "open: the point is shorter than the polynomial's variables");
        reqs.push_back({ScRequest::Open, pevals[i], DevPtr(), lens[i], FrVec(points[i].begin(), points[i].begin() + n), DevPtr()});
    }
    std::vector<ScResult> rounds = be.sumcheck_batch(reqs);  // :309-323, all items at once
    std::map<std::pair<const void *, size_t>, DevPtr> first;
    OpensInFlight out;
    auto cuts = std::make_shared<std::vector<std::vector<size_t>>>();
    for (size_t i = 0; i < lens.size(); ++i) {
        DevPtr q0 = first.emplace(std::make_pair((const void *)pevals[i].get(), lens[i]), rounds[i].out).first->second;
        std::vector<SrsPtr> srs;
        std::vector<DevPtr> bufs;
        std::vector<size_t> ls;
        detail::open_items(pg, rounds[i].out, lens[i], l, srs, bufs, ls);
        if (!bufs.empty()) bufs[0] = q0;
        out.values.push_back(rounds[i].last_f);
        cuts->push_back(q.add(srs, bufs, ls));
        q.keep.push_back(rounds[i].out);  // the q buffers must outlive the batched MSM
    }
    std::vector<Fr> vals = out.values;
    out.finish = [pass = q.ticket(), cuts, vals] {
        std::vector<Opening> res;
        for (size_t i = 0; i < vals.size(); ++i) res.push_back({vals[i], detail::pick(pass, (*cuts)[i])});
        return res;
    };
    return out;
}

// several d_open (dpoly_comm.rs:355-398): local fold rounds and the exchange of the local VALUES happen now; the commitments
// of the local opens and (leader) of the root opens on the gathered values are queued
inline std::function<std::vector<Opening>()> d_open_many_q(Ctx &be, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals, const std::vector<size_t> &lens,
                                                           const std::vector<FrVec> &points, Net &net) {
    size_t k = lens.size(), plog = log2_floor(net.n_parties), np = net.n_parties;
    if (!k) return [] { return std::vector<Opening>{}; };
    std::vector<FrVec> lo, hi;
    for (auto &pt : points) {
        if (pt.size() < plog) throw ZkError(ZK_ERR_INVALID, "d_open: the point is shorter than the party bits");
        hi.emplace_back(pt.begin() + plog, pt.end());
        lo.emplace_back(pt.begin(), pt.begin() + plog);
    }
    auto local = std::make_shared<OpensInFlight>(open_many_q(be, q, pg, pevals, lens, hi));
    std::vector<FrVec> vals = net.all_gather_fr(local->values);  // [party][k]
    auto root = std::make_shared<OpensInFlight>();
    if (net.is_leader()) {
        FrVec tab;  // [k][party]
        for (size_t i = 0; i < k; ++i)
            for (size_t p = 0; p < np; ++p) tab.push_back(vals[p][i]);
        DevPtr d = be.to_device(tab);
        std::vector<DevPtr> roots;
        for (size_t i = 0; i < k; ++i) roots.push_back(d.fr(np * i));
        *root = open_many_q(be, q, pg, roots, std::vector<size_t>(k, np), lo);
        q.keep.push_back(d);
    }
    return [&be, &net, local, root, k, np] {
        std::vector<Opening> loc = local->finish();
        G1Vec flat;
        std::vector<size_t> cuts{0};
        for (auto &o : loc) {
            flat.insert(flat.end(), o.proofs.begin(), o.proofs.end());
            cuts.push_back(flat.size());
        }
        std::vector<G1Vec> prfs = net.all_gather_g1(flat);
        std::vector<Opening> out(k, Opening{Fr::zero(), {}});
        if (!net.is_leader()) return out;
        G1Vec pi = be.g1_lincomb_batch(detail::by_item(prfs), FrVec(np, Fr{{1, 0, 0, 0}}), flat.size());
        std::vector<Opening> roots = root->finish();
        for (size_t i = 0; i < k; ++i) {
            out[i] = roots[i];  // root proofs FIRST (:379-384)
            out[i].proofs.insert(out[i].proofs.end(), pi.begin() + cuts[i], pi.begin() + cuts[i + 1]);
        }
        return out;
    };
}

// several c_open (dpoly_comm.rs:401-464) whose q_i commitments share ONE queued d_msm
inline std::function<std::vector<Opening>()> c_open_many_q(Ctx &be, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals, const std::vector<size_t> &lens,
                                                           const std::vector<FrVec> &points, const PackedSharingParams &pp, Net &net) {
    size_t k = lens.size();
    std::vector<ScRequest> reqs;
    for (size_t i = 0; i < k; ++i) {
        size_t n = Ctx::log2_exact(lens[i]);
        if (points[i].size() < n) throw ZkError(ZK_ERR_INVALID, "c_open: the point is shorter than the polynomial's variables");
        reqs.push_back({ScRequest::Open, pevals[i], DevPtr(), lens[i], FrVec(points[i].begin(), points[i].begin() + n), DevPtr()});
    }
    std::vector<ScResult> rounds = be.sumcheck_batch(reqs);  // :418-432
    std::map<std::pair<const void *, size_t>, DevPtr> first;
    std::vector<SrsPtr> srs;
    std::vector<DevPtr> bufs;
    std::vector<size_t> ms, cuts{0};
    for (size_t i = 0; i < k; ++i) {
        DevPtr q0 = first.emplace(std::make_pair((const void *)pevals[i].get(), lens[i]), rounds[i].out).first->second;
        size_t at = bufs.size();
        detail::open_items(pg, rounds[i].out, lens[i], pp.l, srs, bufs, ms);
        if (bufs.size() > at) bufs[at] = q0;
        q.keep.push_back(rounds[i].out);
        cuts.push_back(ms.size());
    }
    std::function<G1Vec()> f_com = ms.empty() ? std::function<G1Vec()>([] { return G1Vec{}; }) : d_msm_q(be, q, srs, bufs, ms, pp, net);
    // phase 2 (:440-462): pss2ss of the last value, log2(l) more rounds on the l-vector; its MSMs are queued too
    struct Tail {
        Fr value;
        std::vector<size_t> items;
    };
    auto tails = std::make_shared<std::vector<Tail>>();
    for (size_t i = 0; i < k; ++i) {
        FrVec cur = pss2ss(rounds[i].last_f, pp, net);
        Tail t;
        for (size_t r = 0; r < log2_floor(pp.l); ++r) {
            size_t h = cur.size() / 2;
            FrVec qi(h), nx(h);
            for (size_t j = 0; j < h; ++j) {
                qi[j] = cur[j + h] - cur[j];
                nx[j] = cur[j] + points[i].at(r) * (cur[j + h] - cur[j]);
            }
            t.items.push_back(q.add({detail::level_for(pg, h * pp.l)}, {be.to_device(qi)}, {h})[0]);  // :457 (a plain local G::msm)
            cur = nx;
        }
        t.value = cur[0];
        tails->push_back(t);
    }
    return [pass = q.ticket(), f_com, tails, cuts, k] {
        G1Vec com = f_com();
        std::vector<Opening> out;
        for (size_t i = 0; i < k; ++i) {
            Opening o{(*tails)[i].value, G1Vec(com.begin() + cuts[i], com.begin() + cuts[i + 1])};
            for (size_t it : (*tails)[i].items) o.proofs.push_back(pass->at(it));
            out.push_back(o);
        }
        return out;
    };
}

// several independent c_sumcheck_product (dsumcheck.rs:148-285) on tables of one length: their phase-1 loops run as ONE
// batched call, the pss2ss hand-offs (:224-225) and phase 2 follow item by item in the reference's order
inline std::vector<std::vector<Triple>> c_sumcheck_product_many(Ctx &be, const std::vector<std::pair<DevPtr, DevPtr>> &pairs, size_t len, const FrVec &challenge,
                                                                const PackedSharingParams &pp, Net &net) {
    size_t n = Ctx::log2_exact(len);
    std::vector<ScRequest> reqs;
    for (auto &fg : pairs) {
        detail::trace(be, 'c', fg.first, fg.second, len, challenge, n);
        reqs.push_back({ScRequest::Product, fg.first, fg.second, len, FrVec(challenge.begin(), challenge.begin() + n), DevPtr()});
    }
    std::vector<std::vector<Triple>> out;
    for (ScResult &r : be.sumcheck_batch(reqs)) {
        std::vector<Triple> tr = detail::triples_of(r.sums);
        FrVec vf = pss2ss(r.last_f, pp, net), vg = pss2ss(r.last_g, pp, net);
        for (size_t i = 0; i < log2_floor(pp.l); ++i) tr.push_back(detail::round_product(vf, vg, challenge.at(i)));
        tr.push_back({Fr::zero(), vf[0] * vg[0], Fr::zero()});
        out.push_back(tr);
    }
    return out;
}

// several independent d_sumcheck_product (dsumcheck.rs:359-512).  The local phases run NOW as one batched call; the closure
// performs the exchange -- the per-item gathers of the round tuples (:437) as ONE all-gather of the concatenated payloads --
// and the leader rounds (:440-507).  A protocol step calls it after it has started its MSM pass.
struct DsumcheckItem {
    DevPtr f, g;
    size_t len;
    FrVec challenge;
};
inline std::function<std::vector<std::vector<Triple>>()> d_sumcheck_product_many_q(Ctx &be, const std::vector<DsumcheckItem> &items, Net &net) {
    if (items.empty()) return [] { return std::vector<std::vector<Triple>>{}; };
    size_t s = log2_floor(net.n_parties);
    std::vector<ScRequest> reqs;
    auto ns = std::make_shared<std::vector<size_t>>();
    auto chals = std::make_shared<std::vector<FrVec>>();
    for (auto &it : items) {
        size_t n = Ctx::log2_exact(it.len);
        if (it.challenge.size() < n + s) throw ZkError(ZK_ERR_INVALID, "d_sumcheck_product: fewer challenges than local + party rounds");
        detail::trace(be, 'd', it.f, it.g, it.len, it.challenge, n + s);
        ns->push_back(n);
        chals->push_back(it.challenge);
        reqs.push_back({ScRequest::Product, it.f, it.g, it.len, FrVec(it.challenge.begin(), it.challenge.begin() + n), DevPtr()});
    }
    auto phase1 = std::make_shared<std::vector<ScResult>>(be.sumcheck_batch(reqs));
    return [&net, phase1, ns, chals, s] {
        FrVec local;
        std::vector<size_t> cuts{0};
        for (ScResult &r : *phase1) {
            local.insert(local.end(), r.sums.begin(), r.sums.end());
            local.push_back(r.last_g), local.push_back(r.last_f), local.push_back(Fr::zero());  // marker (g, f, 0)  :433
            cuts.push_back(local.size());
        }
        std::vector<FrVec> all = net.all_gather_fr(local);
        std::vector<std::vector<Triple>> out(ns->size());
        if (!net.is_leader()) return out;
        for (size_t k = 0; k < ns->size(); ++k) {
            size_t n = (*ns)[k];
            std::vector<Triple> tr(n, Triple{Fr::zero(), Fr::zero(), Fr::zero()});
            FrVec f, g;
            for (auto &a : all) {
                const Fr *m = &a[cuts[k]];
                for (size_t i = 0; i < n; ++i)
                    for (size_t c = 0; c < 3; ++c) tr[i][c] += m[3 * i + c];
                f.push_back(m[3 * n + 1]);  // :448
                g.push_back(m[3 * n]);      // :449
            }
            for (size_t i = 0; i < s; ++i) tr.push_back(detail::round_product(f, g, (*chals)[k].at(n + i)));
            out[k] = tr;
        }
        return out;
    };
}


// =====================================================================================================================
// One batch for the sumcheck-family kernels of a whole protocol step (round 5).
//
// The `*_q` forms above run their kernels in a batch of their own and block for it: the wiring step of a proof is then ~10
// blocking calls (2.c, the opens of V, the 57 d_sumcheck_products, the 57 d_opens, the leader's top tree, the opens of step 4)
// of which most are latency chains that leave the chip nearly empty -- 4.2 ms of a 68 ms proof at n = 20 (hyperplonk --marks).
// None of them needs another one's RESULT on the device, only the tables; so they can all be enqueued first and run as ONE
// zk_sumcheck_batch (ScQueue), after which the host parts follow in the reference's order.  The `*_sq` forms are the `*_q`
// forms cut in two: phase A adds the requests (the output buffers of the opens are allocated now, so that their quotient
// commitments can be queued in the MsmQueue at once) and returns phase B, which is called after ScQueue::run(): the exchanges that
// need a kernel result (pss2ss of the last values, the gather of the local open values, the leader's root opens) and the MSM
// items that depend on them.  Phase B returns the finishing closure of the `*_q` forms.  Same outputs, bit for bit.
// =====================================================================================================================
struct ScPass {
    std::vector<ScResult> res;
    bool done = false;
    const ScResult &at(size_t i) const {
        if (!done) throw ZkError(ZK_ERR_INVALID, "ScQueue: a result was read before the batch ran");
        return res.at(i);
    }
};
using ScPassRef = std::shared_ptr<ScPass>;

class ScQueue {
  public:
    explicit ScQueue(Ctx &be) : be_(be), pass_(std::make_shared<ScPass>()) {}
    // -> index of the request's result in the pass; Open / Fold outputs are allocated here (a later MSM item may name them)
    size_t add(ScRequest r) {
        size_t n = Ctx::log2_exact(r.len);
        if (r.kind == ScRequest::Open && !r.out) r.out = be_.alloc_fr(r.len - 1);
        if (r.kind == ScRequest::Fold && !r.out) r.out = be_.alloc_fr(r.len >> std::min(n, r.chal.size()));
        reqs_.push_back(std::move(r));
        return reqs_.size() - 1;
    }
    const DevPtr &out_of(size_t i) const { return reqs_.at(i).out; }
    ScPassRef ticket() const { return pass_; }
    bool empty() const { return reqs_.empty(); }
    void run() {
        pass_->res = be_.sumcheck_batch(reqs_);
        pass_->done = true;
        pass_ = std::make_shared<ScPass>();
        reqs_.clear();
    }

  private:
    Ctx &be_;
    std::vector<ScRequest> reqs_;
    ScPassRef pass_;
};

template <class T>
using AfterBatch = std::function<T()>;  // phase B: call after ScQueue::run()

// several c_sumcheck_product on tables of one length (c_sumcheck_product_many): A adds, B = the pss2ss hand-offs and phase 2
inline AfterBatch<std::vector<std::vector<Triple>>> c_sumcheck_product_many_sq(Ctx &be, ScQueue &sq, const std::vector<std::pair<DevPtr, DevPtr>> &pairs, size_t len,
                                                                               const FrVec &challenge, const PackedSharingParams &pp, Net &net) {
    size_t n = Ctx::log2_exact(len);
    std::vector<size_t> idx;
    for (auto &fg : pairs) {
        detail::trace(be, 'c', fg.first, fg.second, len, challenge, n);
        idx.push_back(sq.add({ScRequest::Product, fg.first, fg.second, len, FrVec(challenge.begin(), challenge.begin() + n), DevPtr()}));
    }
    return [pass = sq.ticket(), idx, challenge, &pp, &net] {
        std::vector<std::vector<Triple>> out;
        for (size_t i : idx) {
            const ScResult &r = pass->at(i);
            std::vector<Triple> tr = detail::triples_of(r.sums);
            FrVec vf = pss2ss(r.last_f, pp, net), vg = pss2ss(r.last_g, pp, net);
            for (size_t k = 0; k < log2_floor(pp.l); ++k) tr.push_back(detail::round_product(vf, vg, challenge.at(k)));
            tr.push_back({Fr::zero(), vf[0] * vg[0], Fr::zero()});
            out.push_back(tr);
        }
        return out;
    };
}

// sumcheck_product (dsumcheck.rs:28-90) with its kernels in the batch
inline AfterBatch<std::vector<Triple>> sumcheck_product_sq(Ctx &be, ScQueue &sq, const DevPtr &f, const DevPtr &g, size_t len, const FrVec &challenge) {
    size_t n = Ctx::log2_exact(len);
    detail::trace(be, 'p', f, g, len, challenge, n);
    size_t i = sq.add({ScRequest::Product, f, g, len, FrVec(challenge.begin(), challenge.begin() + n), DevPtr()});
    return [pass = sq.ticket(), i] {
        const ScResult &r = pass->at(i);
        std::vector<Triple> out = detail::triples_of(r.sums);
        out.push_back({Fr::zero(), r.last_f * r.last_g, Fr::zero()});
        return out;
    };
}

// open_many_q: B has nothing to exchange -- the values are read when the finishing closure runs
inline OpensInFlight open_many_sq(ScQueue &sq, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals, const std::vector<size_t> &lens,
                                  const std::vector<FrVec> &points, size_t l = 1) {
    std::map<std::pair<const void *, size_t>, DevPtr> first;
    auto cuts = std::make_shared<std::vector<std::vector<size_t>>>();
    auto idx = std::make_shared<std::vector<size_t>>();
    for (size_t i = 0; i < lens.size(); ++i) {
        size_t n = Ctx::log2_exact(lens[i]);
        if (points[i].size() < n) throw ZkError(ZK_ERR_INVALID, "open: the point is shorter than the polynomial's variables");
        size_t k = sq.add({ScRequest::Open, pevals[i], DevPtr(), lens[i], FrVec(points[i].begin(), points[i].begin() + n), DevPtr()});
        idx->push_back(k);
        const DevPtr &qbuf = sq.out_of(k);
        DevPtr q0 = first.emplace(std::make_pair((const void *)pevals[i].get(), lens[i]), qbuf).first->second;
        std::vector<SrsPtr> srs;
        std::vector<DevPtr> bufs;
        std::vector<size_t> ls;
        detail::open_items(pg, qbuf, lens[i], l, srs, bufs, ls);
        if (!bufs.empty()) bufs[0] = q0;
        cuts->push_back(q.add(srs, bufs, ls));
        q.keep.push_back(qbuf);
    }
    OpensInFlight out;  // (values: filled by the caller's phase B where an exchange needs them; the closure reads the batch itself)
    out.finish = [mpass = q.ticket(), spass = sq.ticket(), cuts, idx] {
        std::vector<Opening> res;
        for (size_t i = 0; i < idx->size(); ++i) res.push_back({spass->at((*idx)[i]).last_f, detail::pick(mpass, (*cuts)[i])});
        return res;
    };
    return out;
}

// d_open_many_q in two halves: A = the local opens' kernels and quotient commitments; B = the gather of the local values and, on the
// leader, the root opens (their tables are 8 l elements: a batch of their own, run inside B)
inline AfterBatch<std::function<std::vector<Opening>()>> d_open_many_sq(Ctx &be, ScQueue &sq, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals,
                                                                        const std::vector<size_t> &lens, const std::vector<FrVec> &points, Net &net) {
    size_t k = lens.size(), plog = log2_floor(net.n_parties), np = net.n_parties;
    if (!k) return [] { return std::function<std::vector<Opening>()>([] { return std::vector<Opening>{}; }); };
    std::vector<FrVec> lo, hi;
    for (auto &pt : points) {
        if (pt.size() < plog) throw ZkError(ZK_ERR_INVALID, "d_open: the point is shorter than the party bits");
        hi.emplace_back(pt.begin() + plog, pt.end());
        lo.emplace_back(pt.begin(), pt.begin() + plog);
    }
    // (the local opens are added one by one so that their batch indices are known for the gather of the values)
    auto idx = std::make_shared<std::vector<size_t>>();
    auto cuts = std::make_shared<std::vector<std::vector<size_t>>>();
    {
        std::map<std::pair<const void *, size_t>, DevPtr> first;
        for (size_t i = 0; i < k; ++i) {
            size_t n = Ctx::log2_exact(lens[i]);
            if (hi[i].size() < n) throw ZkError(ZK_ERR_INVALID, "open: the point is shorter than the polynomial's variables");
            size_t j = sq.add({ScRequest::Open, pevals[i], DevPtr(), lens[i], FrVec(hi[i].begin(), hi[i].begin() + n), DevPtr()});
            idx->push_back(j);
            const DevPtr &qbuf = sq.out_of(j);
            DevPtr q0 = first.emplace(std::make_pair((const void *)pevals[i].get(), lens[i]), qbuf).first->second;
            std::vector<SrsPtr> srs;
            std::vector<DevPtr> bufs;
            std::vector<size_t> ls;
            detail::open_items(pg, qbuf, lens[i], 1, srs, bufs, ls);
            if (!bufs.empty()) bufs[0] = q0;
            cuts->push_back(q.add(srs, bufs, ls));
            q.keep.push_back(qbuf);
        }
    }
    ScPassRef spass = sq.ticket();
    MsmPassRef mpass = q.ticket();
    return [&be, &q, &net, &pg, spass, mpass, idx, cuts, lo, k, np]() -> std::function<std::vector<Opening>()> {
        FrVec mine;
        for (size_t j : *idx) mine.push_back(spass->at(j).last_f);
        std::vector<FrVec> vals = net.all_gather_fr(mine);  // [party][k]
        auto root = std::make_shared<OpensInFlight>();
        if (net.is_leader()) {
            // the root opens (dpoly_comm.rs:372-378: `open` of the N_p gathered values) are tables of 8 l elements: their fold rounds run
            // here on the host -- the same q = hi - lo, lo + r (hi - lo) -- and only the quotients go to the device for their commitments
            // (as a batch of their own the k of them were a blocking launch chain of ~0.3 ms)
            FrVec qall;
            auto rvals = std::make_shared<std::vector<Fr>>(k);
            for (size_t i = 0; i < k; ++i) {
                FrVec cur;
                for (size_t p = 0; p < np; ++p) cur.push_back(vals[p][i]);
                for (size_t r = 0; cur.size() > 1; ++r) {
                    size_t h = cur.size() / 2;
                    FrVec nx(h);
                    for (size_t j = 0; j < h; ++j) {
                        Fr d = cur[j + h] - cur[j];
                        qall.push_back(d);
                        nx[j] = cur[j] + lo[i].at(r) * d;
                    }
                    cur = nx;
                }
                (*rvals)[i] = cur[0];
            }
            auto rcuts = std::make_shared<std::vector<std::vector<size_t>>>();
            if (np > 1) {
                DevPtr dq = be.to_device(qall);  // [k][N_p - 1]
                q.keep.push_back(dq);
                for (size_t i = 0; i < k; ++i) {
                    std::vector<SrsPtr> srs;
                    std::vector<DevPtr> bufs;
                    std::vector<size_t> ls;
                    detail::open_items(pg, dq.fr((np - 1) * i), np, 1, srs, bufs, ls);
                    rcuts->push_back(q.add(srs, bufs, ls));
                }
            } else {
                rcuts->assign(k, {});
            }
            root->finish = [rpass = q.ticket(), rcuts, rvals, k] {
                std::vector<Opening> res;
                for (size_t i = 0; i < k; ++i) res.push_back({(*rvals)[i], detail::pick(rpass, (*rcuts)[i])});
                return res;
            };
        }
        return [&be, &net, spass, mpass, idx, cuts, root, k, np] {
            G1Vec flat;
            std::vector<size_t> ends{0};
            for (size_t i = 0; i < k; ++i) {
                G1Vec pr = detail::pick(mpass, (*cuts)[i]);
                flat.insert(flat.end(), pr.begin(), pr.end());
                ends.push_back(flat.size());
            }
            std::vector<G1Vec> prfs = net.all_gather_g1(flat);
            std::vector<Opening> out(k, Opening{Fr::zero(), {}});
            if (!net.is_leader()) return out;
            G1Vec pi = be.g1_lincomb_batch(detail::by_item(prfs), FrVec(np, Fr{{1, 0, 0, 0}}), flat.size());
            std::vector<Opening> roots = root->finish();
            for (size_t i = 0; i < k; ++i) {
                out[i] = roots[i];  // root proofs FIRST (:379-384)
                out[i].proofs.insert(out[i].proofs.end(), pi.begin() + ends[i], pi.begin() + ends[i + 1]);
            }
            return out;
        };
    };
}

// c_open_many_q in two halves: A = fold rounds + the ONE queued d_msm over all quotients; B = pss2ss of the last values and the
// log2(l) extra rounds, whose small MSMs are queued too
inline AfterBatch<std::function<std::vector<Opening>()>> c_open_many_sq(Ctx &be, ScQueue &sq, MsmQueue &q, const PowersOfG &pg, const std::vector<DevPtr> &pevals,
                                                                        const std::vector<size_t> &lens, const std::vector<FrVec> &points,
                                                                        const PackedSharingParams &pp, Net &net) {
    size_t k = lens.size();
    std::map<std::pair<const void *, size_t>, DevPtr> first;
    std::map<std::tuple<const void *, size_t, std::vector<uint64_t>>, size_t> same;
    std::vector<SrsPtr> srs;
    std::vector<DevPtr> bufs;
    std::vector<size_t> ms;
    auto cuts = std::make_shared<std::vector<size_t>>(1, 0);
    auto idx = std::make_shared<std::vector<size_t>>();
    for (size_t i = 0; i < k; ++i) {
        size_t n = Ctx::log2_exact(lens[i]);
        if (points[i].size() < n) throw ZkError(ZK_ERR_INVALID, "c_open: the point is shorter than the polynomial's variables");
        // (the same table at the same point again -- cpermcheck opens num / den twice, dhyperplonk.rs:1324 and :1371 -- is the same
        // request: its kernels run once, and its MSM items name the same quotient buffers, which the MsmQueue computes once)
        FrVec pt(points[i].begin(), points[i].begin() + n);
        std::vector<uint64_t> raw;
        for (const Fr &x : pt) raw.insert(raw.end(), x.v, x.v + 4);
        auto same_key = std::make_tuple((const void *)pevals[i].get(), lens[i], raw);
        auto seen = same.find(same_key);
        size_t j = seen != same.end() ? seen->second : sq.add({ScRequest::Open, pevals[i], DevPtr(), lens[i], pt, DevPtr()});
        same.emplace(same_key, j);
        idx->push_back(j);
        const DevPtr &qbuf = sq.out_of(j);
        DevPtr q0 = first.emplace(std::make_pair((const void *)pevals[i].get(), lens[i]), qbuf).first->second;
        size_t at = bufs.size();
        detail::open_items(pg, qbuf, lens[i], pp.l, srs, bufs, ms);
        if (bufs.size() > at) bufs[at] = q0;
        q.keep.push_back(qbuf);
        cuts->push_back(ms.size());
    }
    ScPassRef spass = sq.ticket();
    std::vector<FrVec> pts = points;
    return [&be, &q, &net, &pp, &pg, spass, idx, cuts, srs, bufs, ms, pts, k]() -> std::function<std::vector<Opening>()> {
        // (the d_msm over the quotients is queued HERE, after the batch: with real parties d_msm_q pre-scales its scalars by lambda_p on
        // the device when it is called, and the quotient buffers are filled by the batch)
        std::function<G1Vec()> f_com = ms.empty() ? std::function<G1Vec()>([] { return G1Vec{}; }) : d_msm_q(be, q, srs, bufs, ms, pp, net);
        struct Tail {
            Fr value;
            std::vector<size_t> items;
        };
        auto tails = std::make_shared<std::vector<Tail>>();
        for (size_t i = 0; i < k; ++i) {
            FrVec cur = pss2ss(spass->at((*idx)[i]).last_f, pp, net);
            Tail t;
            for (size_t r = 0; r < log2_floor(pp.l); ++r) {
                size_t h = cur.size() / 2;
                FrVec qi(h), nx(h);
                for (size_t j = 0; j < h; ++j) {
                    qi[j] = cur[j + h] - cur[j];
                    nx[j] = cur[j] + pts[i].at(r) * (cur[j + h] - cur[j]);
                }
                t.items.push_back(q.add({detail::level_for(pg, h * pp.l)}, {be.to_device(qi)}, {h})[0]);  // :457 (a plain local G::msm)
                cur = nx;
            }
            t.value = cur[0];
            tails->push_back(t);
        }
        return [mpass = q.ticket(), f_com, tails, cuts, k] {
            G1Vec com = f_com();
            std::vector<Opening> out;
            for (size_t i = 0; i < k; ++i) {
                Opening o{(*tails)[i].value, G1Vec(com.begin() + (*cuts)[i], com.begin() + (*cuts)[i + 1])};
                for (size_t it : (*tails)[i].items) o.proofs.push_back(mpass->at(it));
                out.push_back(o);
            }
            return out;
        };
    };
}

// d_sumcheck_product_many_q with its local phases in the batch: the returned closure (exchange + leader rounds) is called when
// the transcript is assembled, long after ScQueue::run()
inline std::function<std::vector<std::vector<Triple>>()> d_sumcheck_product_many_sq(Ctx &be, ScQueue &sq, const std::vector<DsumcheckItem> &items, Net &net) {
    if (items.empty()) return [] { return std::vector<std::vector<Triple>>{}; };
    size_t s = log2_floor(net.n_parties);
    auto ns = std::make_shared<std::vector<size_t>>();
    auto chals = std::make_shared<std::vector<FrVec>>();
    auto idx = std::make_shared<std::vector<size_t>>();
    for (auto &it : items) {
        size_t n = Ctx::log2_exact(it.len);
        if (it.challenge.size() < n + s) throw ZkError(ZK_ERR_INVALID, "d_sumcheck_product: fewer challenges than local + party rounds");
        detail::trace(be, 'd', it.f, it.g, it.len, it.challenge, n + s);
        ns->push_back(n);
        chals->push_back(it.challenge);
        idx->push_back(sq.add({ScRequest::Product, it.f, it.g, it.len, FrVec(it.challenge.begin(), it.challenge.begin() + n), DevPtr()}));
    }
    return [&net, pass = sq.ticket(), idx, ns, chals, s] {
        FrVec local;
        std::vector<size_t> cuts{0};
        for (size_t i : *idx) {
            const ScResult &r = pass->at(i);
            local.insert(local.end(), r.sums.begin(), r.sums.end());
            local.push_back(r.last_g), local.push_back(r.last_f), local.push_back(Fr::zero());  // marker (g, f, 0)  :433
            cuts.push_back(local.size());
        }
        std::vector<FrVec> all = net.all_gather_fr(local);
        std::vector<std::vector<Triple>> out(ns->size());
        if (!net.is_leader()) return out;
        for (size_t k = 0; k < ns->size(); ++k) {
            size_t n = (*ns)[k];
            std::vector<Triple> tr(n, Triple{Fr::zero(), Fr::zero(), Fr::zero()});
            FrVec f, g;
            for (auto &a : all) {
                const Fr *m = &a[cuts[k]];
                for (size_t i = 0; i < n; ++i)
                    for (size_t c = 0; c < 3; ++c) tr[i][c] += m[3 * i + c];
                f.push_back(m[3 * n + 1]);  // :448
                g.push_back(m[3 * n]);      // :449
            }
            for (size_t i = 0; i < s; ++i) tr.push_back(detail::round_product(f, g, (*chals)[k].at(n + i)));
            out[k] = tr;
        }
        return out;
    };
}

}  // namespace zkhost
