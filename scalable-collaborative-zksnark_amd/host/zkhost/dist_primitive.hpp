// C++ host mirror of the reference's `dist-primitive` crate: the same function names, argument meaning, output shape and
// ordering, with every loop body executed by libzkhip.so on the GPU through the C ABI
// (include/zkhip.h).  The reference is Rust; no Rust toolchain exists in this image, so this is the compiled host side a
// user of `dist-primitive` switches to (rust/*.rs shows the same calls as the `extern "C"` binding a Rust build would use).
//
// Tables (`&Vec<F>`) are device-resident (DevPtr + length); SRS levels are `Srs` handles; field elements cross the API as
// `Fr` (Montgomery limbs, the reference's memory layout) and points as normalised Jacobian `G1`.  `be` is the compute
// backend (one ctx = one GPU = one party), `net` the party exchange.  What stays on the host is what the reference's LEADER
// does on a handful of elements (phase 2 of the c_ sumchecks, the last s rounds of the d_ ones, the top of the product tree).
//
// This file never reads the oracle and has no CPU fallback: without the library's device code every call throws ZkError.
#pragma once
#include <array>
#include <optional>
#include <utility>

#include "device.hpp"
#include "net.hpp"
#include "pss.hpp"

namespace zkhost {

using Pair = std::array<Fr, 2>;
using Triple = std::array<Fr, 3>;
using PowersOfG = std::vector<SrsPtr>;  // PolynomialCommitment::powers_of_g: level k holds 2^k points (dpoly_comm.rs:18-28)

constexpr size_t NTT_FROM_N = 64;  // party counts from which the PSS maps on tables run as transforms (zk_fr_ntt_map)

inline size_t log2_floor(size_t x) {
    size_t n = 0;
    while (x >> (n + 1)) ++n;
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// utils/operator.rs:23-36
// ---------------------------------------------------------------------------------------------------------------------
template <class T>
std::vector<std::vector<T>> transpose(const std::vector<std::vector<T>> &m) {
    if (m.empty()) throw ZkError(ZK_ERR_INVALID, "transpose: empty matrix (operator.rs:24 asserts)");
    size_t cols = m[0].size();
    std::vector<std::vector<T>> out(cols);
    for (auto &row : m) {
        if (row.size() != cols) throw ZkError(ZK_ERR_INVALID, "transpose: ragged matrix");
        for (size_t c = 0; c < cols; ++c) out[c].push_back(row[c]);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// pss2ss (unpack.rs:72-97): gather 1 Fr, unpack, pack_single every secret, scatter -> this party's Vec<F> of length l
// ---------------------------------------------------------------------------------------------------------------------
inline FrVec pss2ss(const Fr &share, const PackedSharingParams &pp, Net &net) {
    std::vector<FrVec> got = net.all_gather_fr(FrVec{share});
    FrVec shares;
    for (auto &g : got) shares.push_back(g[0]);
    FrVec secrets = pp.unpack(shares), out;
    const Fr &w = pp.pack_single_of_one()[net.party_id];  // pack_single(s)[p] = s * pack_single(1)[p]
    for (const Fr &s : secrets) out.push_back(s * w);
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// d_msm (dmsm.rs:9-43)
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
inline FrVec canonical(const FrVec &m) {
    FrVec c;
    for (const Fr &x : m) c.push_back(x.to_canonical());
    return c;
}
// gathered[party][item] -> the [item][party] order zk_g1_lincomb_batch reads
inline G1Vec by_item(const std::vector<G1Vec> &gathered) {
    G1Vec flat;
    size_t k = gathered.empty() ? 0 : gathered[0].size();
    for (size_t i = 0; i < k; ++i)
        for (auto &g : gathered) flat.push_back(g[i]);
    return flat;
}
inline std::vector<const Srs *> raw(const std::vector<SrsPtr> &v) {
    std::vector<const Srs *> r;
    for (auto &s : v) r.push_back(s.get());
    return r;
}
}  // namespace detail

// bases[k]: a device-resident level, scalars[k]: lens[k] Fr shares in HBM.  Returns this party's share of every result.
// Local: G::msm per batch item (dmsm.rs:19-24), all items in one pipeline pass.  Exchange: the leader closure
// unpack2 -> sum -> pack_from_public([sum; l]) (:29-40) is the public linear map
//     out_p = c_p sum_i lambda_i C_i,   lambda_i = sum_j unpack2[j][i],  c_p = sum_j pack[p][j].
// prescale folds lambda_p into this party's SCALARS before its MSM (MSM(b, lambda s) = lambda MSM(b, s)): the exchange is
// then an all-gather, n - 1 point additions and ONE scalar multiplication by c_p.  Same group element, same output bits.
inline G1Vec d_msm(Ctx &be, const std::vector<SrsPtr> &bases, const std::vector<DevPtr> &scalars, const std::vector<size_t> &lens,
                   const PackedSharingParams &pp, Net &net, bool prescale = true) {
    if (bases.size() != scalars.size() || bases.size() != lens.size()) throw ZkError(ZK_ERR_INVALID, "d_msm: bases / scalars batch sizes differ (dmsm.rs:16)");
    size_t k = lens.size(), p = net.party_id, n = net.n_parties;
    if (!k) return {};
    auto srs = detail::raw(bases);
    if (!prescale || net.echo) {
        // (the no-`comm` echo fabricates the other parties' messages from the local one: apply the map exactly where the reference does)
        std::vector<G1Vec> got = net.all_gather_g1(be.msm_g1_batch(srs, scalars, lens));
        return be.g1_lincomb_batch(detail::by_item(got), detail::canonical(pp.dmsm_coeffs(p)), k);
    }
    Fr lam = pp.lambda(p), cp = pp.c(p);
    if (auto *rn = dynamic_cast<RcclNet *>(&net); rn && rn->owns(be)) {
        // the communicator lives in the same ctx: the whole of d_msm is ONE C-ABI call
        FrVec coeffs(n, cp.to_canonical());
        rn->account(144 * k);  // (the all-gather of the k results happens inside zk_d_msm)
        return be.d_msm(srs, scalars, lens, &lam, coeffs);
    }
    std::vector<DevPtr> scaled;
    for (size_t i = 0; i < k; ++i) scaled.push_back(be.fr_scale(scalars[i], lam, lens[i]));
    std::vector<G1Vec> got = net.all_gather_g1(be.msm_g1_batch(srs, scaled, lens));
    G1Vec sums = be.g1_lincomb_batch(detail::by_item(got), FrVec(n, Fr{{1, 0, 0, 0}}), k);
    return be.g1_lincomb_batch(sums, FrVec{cp.to_canonical()}, k);
}

// ---------------------------------------------------------------------------------------------------------------------
// sumcheck family (dsumcheck.rs)
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
// one leader round on a host vector: (sum lo, sum hi), fold with r
inline Pair round_plain(FrVec &f, const Fr &r) {
    size_t h = f.size() / 2;
    Pair s{Fr::zero(), Fr::zero()};
    FrVec nx(h);
    for (size_t j = 0; j < h; ++j) {
        s[0] += f[j], s[1] += f[j + h];
        nx[j] = f[j] + r * (f[j + h] - f[j]);
    }
    f = nx;
    return s;
}
inline Triple round_product(FrVec &f, FrVec &g, const Fr &r) {
    size_t h = f.size() / 2;
    Triple t{Fr::zero(), Fr::zero(), Fr::zero()};
    FrVec nf(h), ng(h);
    for (size_t j = 0; j < h; ++j) {
        t[0] += f[j] * g[j];
        t[1] += f[j + h] * g[j + h];
        t[2] += (f[j + h] + f[j + h] - f[j]) * (g[j + h] + g[j + h] - g[j]);
        nf[j] = f[j] + r * (f[j + h] - f[j]);
        ng[j] = g[j] + r * (g[j + h] - g[j]);
    }
    f = nf, g = ng;
    return t;
}
inline std::vector<Pair> pairs_of(const FrVec &s) {
    std::vector<Pair> out(s.size() / 2);
    for (size_t i = 0; i < out.size(); ++i) out[i] = {s[2 * i], s[2 * i + 1]};
    return out;
}
// the self-check hook (device.hpp ScTrace): record the operands of a product sumcheck
inline void trace(Ctx &be, char kind, const DevPtr &f, const DevPtr &g, size_t len, const FrVec &challenge, size_t count) {
    if (be.sc_trace) be.sc_trace->push_back({kind, f, g, len, FrVec(challenge.begin(), challenge.begin() + std::min(count, challenge.size()))});
}
inline std::vector<Triple> triples_of(const FrVec &s) {
    std::vector<Triple> out(s.size() / 3);
    for (size_t i = 0; i < out.size(); ++i) out[i] = {s[3 * i], s[3 * i + 1], s[3 * i + 2]};
    return out;
}
}  // namespace detail

// dsumcheck.rs:6-26 -> n + 1 pairs, the last one (0, last)
inline std::vector<Pair> sumcheck(Ctx &be, const DevPtr &evaluation, size_t len, const FrVec &challenge) {
    ScResult r = be.sumcheck(evaluation, len, challenge);
    std::vector<Pair> out = detail::pairs_of(r.sums);
    out.push_back({Fr::zero(), r.last_f});
    return out;
}

// dsumcheck.rs:28-90 -> n + 1 triples, the last one (0, f g, 0)
inline std::vector<Triple> sumcheck_product(Ctx &be, const DevPtr &f, const DevPtr &g, size_t len, const FrVec &challenge) {
    detail::trace(be, 'p', f, g, len, challenge, Ctx::log2_exact(len));
    ScResult r = be.sumcheck_product(f, g, len, challenge);
    std::vector<Triple> out = detail::triples_of(r.sums);
    out.push_back({Fr::zero(), r.last_f * r.last_g, Fr::zero()});
    return out;
}

// dsumcheck.rs:92-146 -> n + log2(l) + 1 pairs; phase 2 re-uses challenge[0 .. log2 l) (:129)
inline std::vector<Pair> c_sumcheck(Ctx &be, const DevPtr &shares, size_t len, const FrVec &challenge, const PackedSharingParams &pp, Net &net) {
    ScResult r = be.sumcheck(shares, len, challenge);
    std::vector<Pair> out = detail::pairs_of(r.sums);
    FrVec v = pss2ss(r.last_f, pp, net);
    for (size_t i = 0; i < log2_floor(pp.l); ++i) out.push_back(detail::round_plain(v, challenge.at(i)));
    out.push_back({Fr::zero(), v[0]});
    return out;
}

// dsumcheck.rs:148-285 -> n + log2(l) + 1 triples
inline std::vector<Triple> c_sumcheck_product(Ctx &be, const DevPtr &shares_f, const DevPtr &shares_g, size_t len, const FrVec &challenge,
                                              const PackedSharingParams &pp, Net &net) {
    detail::trace(be, 'c', shares_f, shares_g, len, challenge, Ctx::log2_exact(len));
    ScResult r = be.sumcheck_product(shares_f, shares_g, len, challenge);
    std::vector<Triple> out = detail::triples_of(r.sums);
    FrVec vf = pss2ss(r.last_f, pp, net);  // :224
    FrVec vg = pss2ss(r.last_g, pp, net);  // :225
    for (size_t i = 0; i < log2_floor(pp.l); ++i) out.push_back(detail::round_product(vf, vg, challenge.at(i)));
    out.push_back({Fr::zero(), vf[0] * vg[0], Fr::zero()});  // :282
    return out;
}

// dsumcheck.rs:287-357.  Leader: n' + s pairs (s = log2 parties); workers: empty
inline std::vector<Pair> d_sumcheck(Ctx &be, const DevPtr &partial_poly, size_t len, const FrVec &challenge, Net &net) {
    size_t n = Ctx::log2_exact(len), s = log2_floor(net.n_parties);
    ScResult r = be.sumcheck(partial_poly, len, challenge);
    FrVec local = r.sums;
    local.push_back(Fr::zero());
    local.push_back(r.last_f);
    std::vector<FrVec> all = net.all_gather_fr(local);
    if (!net.is_leader()) return {};
    std::vector<Pair> out(n, Pair{Fr::zero(), Fr::zero()});
    FrVec v;
    for (auto &a : all) {  // per-round sums over the parties (:440-447 pattern)
        for (size_t i = 0; i < n; ++i) out[i][0] += a[2 * i], out[i][1] += a[2 * i + 1];
        v.push_back(a[2 * n + 1]);
    }
    for (size_t i = 0; i < s; ++i) out.push_back(detail::round_plain(v, challenge.at(n + i)));
    return out;
}

// dsumcheck.rs:359-512.  Leader: n' + s triples; workers: empty.  The marker tuple a party sends is (g, f, 0) (:433)
inline std::vector<Triple> d_sumcheck_product(Ctx &be, const DevPtr &partial_f, const DevPtr &partial_g, size_t len, const FrVec &challenge, Net &net) {
    size_t n = Ctx::log2_exact(len), s = log2_floor(net.n_parties);
    detail::trace(be, 'd', partial_f, partial_g, len, challenge, n + s);
    ScResult r = be.sumcheck_product(partial_f, partial_g, len, challenge);
    FrVec local = r.sums;
    local.push_back(r.last_g);
    local.push_back(r.last_f);
    local.push_back(Fr::zero());
    std::vector<FrVec> all = net.all_gather_fr(local);
    if (!net.is_leader()) return {};
    std::vector<Triple> out(n, Triple{Fr::zero(), Fr::zero(), Fr::zero()});
    FrVec f, g;
    for (auto &a : all) {
        for (size_t i = 0; i < n; ++i)
            for (size_t c = 0; c < 3; ++c) out[i][c] += a[3 * i + c];
        f.push_back(a[3 * n + 1]);  // :448
        g.push_back(a[3 * n]);      // :449
    }
    for (size_t i = 0; i < s; ++i) out.push_back(detail::round_product(f, g, challenge.at(n + i)));
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// product accumulation (dacc_product.rs)
// ---------------------------------------------------------------------------------------------------------------------
// dacc_product.rs:18-23
inline std::pair<size_t, size_t> sub_index(size_t i) {
    size_t x = (i & ~(size_t(1) << log2_floor(i))) << 1;
    return {x, x + 1};
}

// dacc_product.rs:30-57 -> the tree of 2N Fr in HBM: v(x,0) = tree[0::2], v(x,1) = tree[1::2], v(1,x) = tree[N..]
struct ProductTree {
    DevPtr tree;
    size_t N;
    // the three views as the reference returns them (device buffers of N Fr; v(1,x) aliases the tree)
    std::array<DevPtr, 3> views(Ctx &be) const {
        auto eo = be.fr_deinterleave(tree, N);
        return {eo.first, eo.second, tree.fr(N)};
    }
};
inline ProductTree acc_product(Ctx &be, const DevPtr &x, size_t N) { return ProductTree{be.product_tree(x, N), N}; }

// dacc_product.rs:365-414 -> (subtree, the leader's top tree of 2 N_p elements)
inline std::pair<ProductTree, std::optional<FrVec>> d_acc_product(Ctx &be, const DevPtr &inputs, size_t N, Net &net) {
    ProductTree sub = acc_product(be, inputs, N);
    FrVec root = be.to_host(sub.tree.fr(2 * N - 1), 1);  // the forced 0 (:381,:390)
    std::vector<FrVec> roots = net.all_gather_fr(root);
    if (!net.is_leader()) return {sub, std::nullopt};
    FrVec t;
    for (auto &r : roots) t.push_back(r[0]);
    size_t np = net.n_parties;
    for (size_t i = np; i < 2 * np - 1; ++i) {
        auto ab = sub_index(i);
        t.push_back(t[ab.first] * t[ab.second]);
    }
    t.push_back(Fr::zero());
    return {sub, t};
}

// dacc_product.rs:296-363: local subtree; every party sends its LAST min(N_p, 2N) entries (:321-329); the leader interleaves
// them level by level (:339-349) and appends N_p - 1 products and a 0
inline std::pair<ProductTree, std::optional<FrVec>> c_acc_product(Ctx &be, const DevPtr &inputs, size_t N, const PackedSharingParams &pp, Net &net) {
    ProductTree sub = acc_product(be, inputs, N);
    size_t np = pp.n, num_to_send = std::min(np, 2 * N);
    FrVec tail = be.to_host(sub.tree.fr(2 * N - num_to_send), num_to_send);
    std::vector<FrVec> recv = net.all_gather_fr(tail);
    if (!net.is_leader()) return {sub, std::nullopt};
    FrVec tree;
    size_t start = 0;
    for (size_t layer = num_to_send / 2; layer > 0; layer >>= 1) {
        for (size_t j = 0; j < np; ++j) tree.insert(tree.end(), recv[j].begin() + start, recv[j].begin() + start + layer);
        start += layer;
    }
    size_t total = num_to_send * np;
    for (size_t i = total - np; i + 1 < total; ++i) {
        auto ab = sub_index(i);
        tree.push_back(tree[ab.first] * tree[ab.second]);
    }
    tree.push_back(Fr::zero());
    return {sub, tree};
}

// dacc_product.rs:416-428: interleave per-party vectors level by level
inline FrVec merge(const std::vector<FrVec> &results) {
    size_t n = results.at(0).size(), num = 1, start = 0;
    while (num < n + 1) num <<= 1;
    num >>= 1;
    FrVec out;
    while (num > 0 && start + num <= n) {  // (num == 0 would spin for ever in the reference: len = 2^k - 1)
        for (auto &r : results) out.insert(out.end(), r.begin() + start, r.begin() + start + num);
        start += num;
        num >>= 1;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// polynomial commitment (dpoly_comm.rs:236-464)
// ---------------------------------------------------------------------------------------------------------------------
struct Opening {
    Fr value;
    G1Vec proofs;
};

// PolynomialCommitmentCub (dpoly_comm.rs:18-28,36-234): the parameter sets; levels are built and kept as affine records in
// HBM, so `mature` (:141-151, projective -> affine) has nothing left to do
struct PolynomialCommitmentCub {
    PowersOfG powers_of_g;
    // :37-67: level k = g^{eq-basis over s_{n-k} .. s_{n-1}}
    static PolynomialCommitmentCub make(Ctx &be, const FrVec &s) { return {be.srs_powers(s)}; }
    // :197-219: a toy single-party parameter set, level i holds max(1, 2^i / l) synthetic points
    static PolynomialCommitmentCub new_single(Ctx &be, size_t len_log_2, const PackedSharingParams &pp, uint64_t seed = 1) {
        PolynomialCommitmentCub c;
        for (size_t i = 0; i <= len_log_2; ++i) c.powers_of_g.push_back(be.srs_generate(seed * 7919 + 2 * i + 1, seed * 104729 + 2 * i + 3, std::max<size_t>(1, (size_t(1) << i) / pp.l)));
        return c;
    }
    // :220-233: levels 0 .. len_log_2 - log2(party_count) of 2^i synthetic points
    static PolynomialCommitmentCub new_random(Ctx &be, size_t len_log_2, size_t party_count, uint64_t seed = 1) {
        PolynomialCommitmentCub c;
        for (size_t i = 0; i + log2_floor(party_count) <= len_log_2; ++i) c.powers_of_g.push_back(be.srs_generate(seed * 6007 + 2 * i + 5, seed * 15485863 + 2 * i + 7, size_t(1) << i));
        return c;
    }
    // :164-194 for ONE party (every GPU builds its own share): level i -> pack_from_public of every l-chunk
    PolynomialCommitmentCub to_packed(Ctx &be, const PackedSharingParams &pp, size_t party) const {
        FrVec row;
        for (size_t j = 0; j < pp.l; ++j) row.push_back(pp.pack_matrix[party][j].to_canonical());
        PolynomialCommitmentCub c;
        for (auto &lv : powers_of_g) c.powers_of_g.push_back(be.srs_to_packed(*lv, row, pp.l));
        return c;
    }
    const PowersOfG &mature() const { return powers_of_g; }
};

namespace detail {
inline const SrsPtr &level_for(const PowersOfG &pg, size_t len) {
    size_t level = Ctx::log2_exact(len);
    if (level >= pg.size()) throw ZkError(ZK_ERR_INVALID, "commit: no parameter level for this length (dpoly_comm.rs:239-240)");
    return pg[level];
}
// the n commitments of one open (:318-321) as MSM items over the quotient buffer q
inline void open_items(const PowersOfG &pg, const DevPtr &q, size_t len, size_t l, std::vector<SrsPtr> &srs, std::vector<DevPtr> &bufs, std::vector<size_t> &lens) {
    size_t off = 0;
    for (size_t m = len; m > 1; m /= 2) {
        size_t h = m / 2;
        srs.push_back(level_for(pg, h * l));
        bufs.push_back(q.fr(off));
        lens.push_back(h);
        off += h;
    }
}
}  // namespace detail

// dpoly_comm.rs:237-243 (= d_local_commit :269-275)
inline G1 commit(Ctx &be, const PowersOfG &pg, const DevPtr &peval, size_t len) { return be.msm_g1(*detail::level_for(pg, len), peval, len); }
inline G1 d_local_commit(Ctx &be, const PowersOfG &pg, const DevPtr &peval, size_t len) { return commit(be, pg, peval, len); }

// dpoly_comm.rs:299-325 (= d_local_open :327-353): the fold rounds, then the n commitments of the q_i -- they are
// independent, so they run as one batched pass (the reference commits them one by one)
inline Opening open(Ctx &be, const PowersOfG &pg, const DevPtr &peval, size_t len, const FrVec &point) {
    ScResult r = be.open_rounds(peval, len, point);
    std::vector<SrsPtr> srs;
    std::vector<DevPtr> bufs;
    std::vector<size_t> lens;
    detail::open_items(pg, r.out, len, 1, srs, bufs, lens);
    return {r.last_f, be.msm_g1_batch(detail::raw(srs), bufs, lens)};
}
inline Opening d_local_open(Ctx &be, const PowersOfG &pg, const DevPtr &peval, size_t len, const FrVec &point) { return open(be, pg, peval, len, point); }

// dpoly_comm.rs:276-297: every party ends with the sum of the local commitments
inline G1 d_commit(Ctx &be, const PowersOfG &pg, const DevPtr &peval, size_t len, Net &net) {
    std::vector<G1Vec> got = net.all_gather_g1(G1Vec{commit(be, pg, peval, len)});
    return be.g1_lincomb_batch(detail::by_item(got), FrVec(net.n_parties, Fr{{1, 0, 0, 0}}), 1)[0];
}

// dpoly_comm.rs:244-267: d_msm with bases_k = powers_of_g[log2(len_k l)] over a batch of share vectors
inline G1Vec c_commit(Ctx &be, const PowersOfG &pg, const std::vector<DevPtr> &pevals, const std::vector<size_t> &lens, const PackedSharingParams &pp, Net &net) {
    std::vector<SrsPtr> bases;
    for (size_t n : lens) bases.push_back(detail::level_for(pg, n * pp.l));  // :256-257
    return d_msm(be, bases, pevals, lens, pp, net);
}

// dpoly_comm.rs:355-398.  Leader: (root value, root proofs (s entries) ++ summed local proofs (n' entries)) -- root proofs
// FIRST (:379-384); workers: (0, [])
inline Opening d_open(Ctx &be, const PowersOfG &pg, const DevPtr &peval, size_t len, const FrVec &point, Net &net) {
    size_t plog = log2_floor(net.n_parties);
    if (point.size() < plog + Ctx::log2_exact(len)) throw ZkError(ZK_ERR_INVALID, "d_open: the point is shorter than the polynomial's variables");
    Opening local = open(be, pg, peval, len, FrVec(point.begin() + plog, point.end()));
    std::vector<FrVec> vals = net.all_gather_fr(FrVec{local.value});
    std::vector<G1Vec> prfs = net.all_gather_g1(local.proofs);
    if (!net.is_leader()) return {Fr::zero(), {}};
    FrVec root;
    for (auto &v : vals) root.push_back(v[0]);
    Opening top = open(be, pg, be.to_device(root), net.n_parties, FrVec(point.begin(), point.begin() + plog));
    G1Vec pi = be.g1_lincomb_batch(detail::by_item(prfs), FrVec(net.n_parties, Fr{{1, 0, 0, 0}}), local.proofs.size());
    top.proofs.insert(top.proofs.end(), pi.begin(), pi.end());
    return top;
}

// dpoly_comm.rs:401-464: n fold rounds producing every q_i, ONE batched d_msm over them (:436), pss2ss of the last value,
// then log2(l) more rounds on the l-vector re-using point[0..] (:452) -> (value, n + log2 l proofs)
inline Opening c_open(Ctx &be, const PowersOfG &pg, const DevPtr &peval, size_t len, const FrVec &point, const PackedSharingParams &pp, Net &net) {
    ScResult r = be.open_rounds(peval, len, point);  // :418-432
    std::vector<SrsPtr> srs;
    std::vector<DevPtr> bufs;
    std::vector<size_t> lens;
    detail::open_items(pg, r.out, len, pp.l, srs, bufs, lens);
    Opening out;
    out.proofs = d_msm(be, srs, bufs, lens, pp, net);
    FrVec cur = pss2ss(r.last_f, pp, net);  // :440
    for (size_t i = 0; i < log2_floor(pp.l); ++i) {
        size_t h = cur.size() / 2;
        FrVec qi(h), nx(h);
        for (size_t j = 0; j < h; ++j) {
            qi[j] = cur[j + h] - cur[j];
            nx[j] = cur[j] + point.at(i) * (cur[j + h] - cur[j]);
        }
        out.proofs.push_back(be.msm_g1(*detail::level_for(pg, h * pp.l), be.to_device(qi), h));  // :457 (a plain local G::msm)
        cur = nx;
    }
    out.value = cur[0];
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// mle.rs
// ---------------------------------------------------------------------------------------------------------------------
// mle.rs:88-105 -> len >> min(n, points) elements in HBM
inline DevPtr fix_variable(Ctx &be, const DevPtr &evaluations, size_t len, const FrVec &points) { return be.fold(evaluations, len, points); }

// mle.rs:51-86: a device buffer while points <= n; beyond that the value after pss2ss and the l-vector rounds, which re-use
// points[0..] (:78)
struct FixedVariable {
    DevPtr table;  // len >> min(n, points) elements
    size_t len;
    std::optional<Fr> value;  // set when points > n
};
inline FixedVariable d_fix_variable(Ctx &be, const DevPtr &shares, size_t len, const FrVec &points, const PackedSharingParams &pp, Net &net) {
    size_t n = Ctx::log2_exact(len), cnt = points.size();
    FixedVariable out;
    out.table = be.fold(shares, len, FrVec(points.begin(), points.begin() + std::min(n, cnt)));
    out.len = len >> std::min(n, cnt);
    if (cnt <= n) return out;
    FrVec cur = pss2ss(be.to_host(out.table, 1)[0], pp, net);
    for (size_t i = 0; i < std::min(cnt - n, log2_floor(pp.l)); ++i) {
        size_t h = cur.size() / 2;
        FrVec nx(h);
        for (size_t j = 0; j < h; ++j) nx[j] = cur[j] + points[i] * (cur[j + h] - cur[j]);
        cur = nx;
    }
    out.value = cur[0];
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// small exchanges: degree reduction and unpacking (degree_reduce.rs, unpack.rs)
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
inline FrVec column(const std::vector<FrVec> &all, size_t k) {
    FrVec c;
    for (auto &a : all) c.push_back(a[k]);
    return c;
}
}  // namespace detail

// degree_reduce.rs:29-41
inline Fr degree_reduce(const Fr &share, const PackedSharingParams &pp, Net &net) {
    FrVec vals = detail::column(net.all_gather_fr(FrVec{share}), 0);
    return pp.pack_from_public(pp.unpack2(vals))[net.party_id];
}

// degree_reduce.rs:10-26 on a device vector of k shares: HBM all-gather + this party's row of the public map pack o unpack2
inline DevPtr degree_reduce_many(Ctx &be, const DevPtr &shares, size_t k, const PackedSharingParams &pp, Net &net) {
    if (!k) return be.alloc_fr(1);
    DevPtr all = net.all_gather_device(be, shares, 32 * k);  // [n][k]
    return be.fr_apply_matrix({pp.degree_reduce_row(net.party_id)}, all, 1, k, k, 1, k);
}
// the same on a host vector
inline FrVec degree_reduce_many(Ctx &be, const FrVec &shares, const PackedSharingParams &pp, Net &net) {
    return be.to_host(degree_reduce_many(be, be.to_device(shares), shares.size(), pp, net), shares.size());
}

// unpack.rs:8-18: every party receives unpack(shares)[0]
inline Fr d_unpack_0(const Fr &share, const PackedSharingParams &pp, Net &net) { return pp.unpack(detail::column(net.all_gather_fr(FrVec{share}), 0))[0]; }
// unpack.rs:20-35: only `receiver` obtains unpack(shares); the others an empty Vec
inline FrVec d_unpack(const Fr &share, size_t receiver, const PackedSharingParams &pp, Net &net) {
    FrVec vals = detail::column(net.all_gather_fr(FrVec{share}), 0);
    return net.party_id == receiver ? pp.unpack(vals) : FrVec{};
}
// unpack.rs:37-52
inline FrVec d_unpack2(const Fr &share, size_t receiver, const PackedSharingParams &pp, Net &net) {
    FrVec vals = detail::column(net.all_gather_fr(FrVec{share}), 0);
    return net.party_id == receiver ? pp.unpack2(vals) : FrVec{};
}

namespace detail {
// unpack2 of every column of a party-major device matrix [n][k] -> k l elements, out[j l + r]   (unpack.rs:55-70)
inline DevPtr unpack2_columns(Ctx &be, const DevPtr &in, size_t k, const PackedSharingParams &pp) {
    if (pp.n >= NTT_FROM_N) return be.fr_ntt_map(pp.ntt_tables(PackedSharingParams::Map::Unpack2), in, 1, k, k, pp.l, 1);
    return be.fr_apply_matrix(pp.unpack2_matrix, in, 1, k, k, pp.l, 1);
}
// `vals.chunks(l).map(pack_from_public)` on a device buffer of `count` Fr -> [n][k] party-major (the send buffer of an
// all-to-all), k = ceil(count / l); a short last chunk is zero-padded exactly as pack_from_public pads
inline DevPtr pack_chunks(Ctx &be, DevPtr vals, size_t count, const PackedSharingParams &pp) {
    size_t l = pp.l, k = (count + l - 1) / l;
    if (count != k * l) {
        DevPtr padded = be.alloc_fr(k * l);
        be.copy_d2d(padded, vals, 32 * count);
        FrVec z(k * l - count, Fr::zero());
        be.upload(padded.fr(count), z.data(), 32 * z.size());
        vals = padded;
    }
    if (pp.n >= NTT_FROM_N) return be.fr_ntt_map(pp.ntt_tables(PackedSharingParams::Map::Pack), vals, l, 1, k, 1, k);
    std::vector<FrVec> m;
    for (auto &row : pp.pack_matrix) m.emplace_back(row.begin(), row.begin() + l);
    return be.fr_apply_matrix(m, vals, l, 1, k, 1, k);
}
// `merge` on a device buffer [n][k] (what every party sent me) -> (buffer, merged length)
inline std::pair<DevPtr, size_t> merge_device(Ctx &be, const DevPtr &in, size_t k, size_t np) {
    size_t num = 1, start = 0, pos = 0;
    while (num < k + 1) num <<= 1;
    num >>= 1;
    DevPtr out = be.alloc_fr(std::max<size_t>(k * np, 1));
    while (num > 0 && start + num <= k) {
        for (size_t q = 0; q < np; ++q) {
            be.copy_d2d(out.fr(pos), in.fr(q * k + start), 32 * num);
            pos += num;
        }
        start += num;
        num >>= 1;
    }
    return {out, pos};
}
}  // namespace detail

// unpack.rs:55-70: the receiver gets transpose(shares).flat_map(unpack2) (k l elements); the others an empty Vec
inline FrVec d_unpack2_many(Ctx &be, const FrVec &share, size_t receiver, const PackedSharingParams &pp, Net &net) {
    std::vector<FrVec> all = net.all_gather_fr(share);
    if (net.party_id != receiver || share.empty()) return {};
    FrVec flat;
    for (auto &a : all) flat.insert(flat.end(), a.begin(), a.end());
    return be.to_host(detail::unpack2_columns(be, be.to_device(flat), share.size(), pp), share.size() * pp.l);
}

// dacc_product.rs:66-292 -> the shares of v(x,0), v(x,1), v(1,x) as (device buffer, length), un-reduced exactly as the
// reference returns them (its three trailing degree_reduce_many calls discard their results).  DEVICE-RESIDENT: mask ->
// all-to-all of the masked blocks -> unpack2 of every column -> product tree -> strided views -> pack_from_public of every
// l-chunk -> all-to-all of the share vectors -> merge -> unmask; only the N_p-sized tails (the leader's top tree, :213-263)
// travel as host vectors.
struct SharedTable {
    DevPtr buf;
    size_t len;
};
inline std::array<SharedTable, 3> c_acc_product_and_share(Ctx &be, const DevPtr &shares, const DevPtr &masks, const DevPtr &unmask0, const DevPtr &unmask1,
                                                          const DevPtr &unmask2, size_t S, const PackedSharingParams &pp, Net &net) {
    size_t N = pp.n, l = pp.l;
    if (S <= N) throw ZkError(ZK_ERR_INVALID, "c_acc_product_and_share: the table must be longer than the party count (dacc_product.rs:82)");
    size_t bs = S / N;
    DevPtr masked = be.fr_mul(shares, masks, S);                                 // :88-92
    DevPtr recv = net.all_to_all_device(be, masked, 32 * bs, Echo::Slot0);       // everyone's i-th block (:94-104)
    DevPtr mx = detail::unpack2_columns(be, recv, bs, pp);                       // unpack.rs:55-70
    size_t mlen = bs * l;
    auto [sub, leader_tree] = c_acc_product(be, mx, mlen, pp, net);
    size_t num_to_send = std::min(N, 2 * mlen), half = (2 * mlen - num_to_send) / 2;
    auto eo = be.fr_deinterleave(sub.tree, half);  // to_share[0::2], to_share[1::2]  (:118-150)
    // v(1,x) = to_share[mlen..]: `.skip(subtree.len() / 2)` past the end of to_share yields nothing
    SharedTable views[3] = {{eo.first, half}, {eo.second, half}, {sub.tree.fr(std::min(mlen, 2 * mlen - num_to_send)), mlen > num_to_send ? mlen - num_to_send : 0}};
    std::pair<DevPtr, size_t> outs[3];
    for (int v = 0; v < 3; ++v) {
        size_t k = (views[v].len + l - 1) / l;
        DevPtr mine = detail::pack_chunks(be, views[v].buf, views[v].len, pp);       // [N][k]
        DevPtr got = net.all_to_all_device(be, mine, 32 * k, Echo::Identity);        // :155-203
        outs[v] = detail::merge_device(be, got, k, N);
    }
    // leader-tree shares (:213-263): note that the v(1,x) share packs the WHOLE leader tree (:243-250)
    size_t ltlen = num_to_send * N, k0 = (ltlen / 2 + l - 1) / l, k2 = (ltlen + l - 1) / l;
    std::vector<FrVec> payload(N, FrVec(2 * k0 + k2, Fr::zero()));
    if (net.is_leader()) {
        const FrVec &lt = *leader_tree;
        FrVec even, odd;
        for (size_t i = 0; i < lt.size(); ++i) (i % 2 ? odd : even).push_back(lt[i]);
        const FrVec *parts[3] = {&even, &odd, &lt};
        size_t offs[3] = {0, k0, 2 * k0};
        for (int v = 0; v < 3; ++v) {
            size_t cnt = parts[v]->size(), k = (cnt + l - 1) / l;
            FrVec packed = be.to_host(detail::pack_chunks(be, be.to_device(*parts[v]), cnt, pp), N * k);  // [N][k]
            for (size_t p = 0; p < N; ++p)
                for (size_t j = 0; j < k; ++j) payload[p][offs[v] + j] = packed[p * k + j];
        }
    }
    FrVec mine = net.all_to_all_fr(payload, Echo::Identity)[0];  // what the leader (party 0) sent to me
    size_t cut[4] = {0, k0, 2 * k0, 2 * k0 + k2};
    const DevPtr *unmask[3] = {&unmask0, &unmask1, &unmask2};
    std::array<SharedTable, 3> res;
    for (int v = 0; v < 3; ++v) {
        size_t shlen = outs[v].second, k = shlen + (cut[v + 1] - cut[v]);
        DevPtr full = be.alloc_fr(std::max<size_t>(k, 1));
        be.copy_d2d(full, outs[v].first, 32 * shlen);
        be.upload(full.fr(shlen), &mine[cut[v]], 32 * (cut[v + 1] - cut[v]));
        res[v] = {be.fr_mul(full, *unmask[v], k), k};  // unmask (:266-275)
    }
    for (auto &r : res) degree_reduce_many(be, r.buf, r.len / N * 2, pp, net);  // :278-285 -- communication only; the reference drops the results too
    return res;
}

}  // namespace zkhost
