// Transcript verifiers of the C++ host: the reference's checkers (dist-primitive/src/dsumcheck.rs:541-588 `check_sumcheck`,
// `check_sumcheck_product`) and the ANCHORED self-check a protocol driver runs on its own output (the compiled counterpart of
// zkhip/verify.py; `hyperplonk --check`).  They read transcripts -- a few hundred field elements -- so the same checks run
// at n = 5 and at n = 24.
//
// WHAT A CHAIN ALONE PROVES.  On tables of 2^18 elements and more the library derives t1 of every round after the first
// from the verifier's own identity (t1_k := p_{k-1}(r_{k-1}) - t0_k, csrc/zk_fr.hip derive_t1), so `t0 + t1 == previous
// target` holds by construction whatever the t0 / t2 kernels produced.  A chain is a real check only with BOTH ENDS pinned by
// values that come from kernels the product sumcheck does not use:
//     claimed   sum_j f_j g_j      zk_fr_mul + the first round of the PLAIN sumcheck (zk_sumcheck)
//     final     f(r) g(r)          two zk_fold
// With both given, a wrong t0 or t2 anywhere moves some round polynomial and the last target misses `final` (except with
// probability ~ 2 rounds / r over the challenges): the sumcheck verifier's own soundness argument.
//
// The operands come from a trace: a Ctx carrying `sc_trace` gets (kind, f, g, len, challenge) of every product sumcheck of
// dist_primitive.hpp / pipeline.hpp appended (kinds: 'p' sumcheck_product, 'c' c_sumcheck_product, 'd' d_sumcheck_product).
#pragma once
#include <map>
#include <string>

#include "hyperplonk.hpp"

namespace zkhost {

// ---------------------------------------------------------------------------------------------------------------------
// dsumcheck.rs:541-588
// ---------------------------------------------------------------------------------------------------------------------
// :541-555.  `rounds` rows are checked (the reference's N); a further (0, last) row is the evaluation the reference's
// comment leaves out ("Now check the oracle query") and must equal the last target.
inline bool check_sumcheck(const Fr &h, const std::vector<Pair> &proof, const FrVec &challenge, size_t rounds) {
    if (proof.size() < rounds || challenge.size() + 1 < rounds) return false;
    if (!rounds) return true;
    if (proof[0][0] + proof[0][1] != h) return false;
    for (size_t i = 1; i < rounds; ++i) {
        Fr target = (proof[i - 1][1] - proof[i - 1][0]) * challenge[i - 1] + proof[i - 1][0];
        if (proof[i][0] + proof[i][1] != target) return false;
    }
    if (proof.size() > rounds && challenge.size() >= rounds) {
        Fr target = (proof[rounds - 1][1] - proof[rounds - 1][0]) * challenge[rounds - 1] + proof[rounds - 1][0];
        if (!proof[rounds][0].is_zero() || proof[rounds][1] != target) return false;
    }
    return true;
}
// :562-575: the quadratic through (0, t0), (1, t1), (2, t2) at x
inline Fr product_round_target(const Triple &t, const Fr &x) {
    static const Fr half = Fr::from_u64(2).inverse(), two = Fr::from_u64(2), three = Fr::from_u64(3), four = Fr::from_u64(4);
    Fr c = t[0], b = (-t[2] + t[1] * four - t[0] * three) * half, a = (t[2] - t[1] * two + t[0]) * half;
    return a * x * x + b * x + c;
}
// :558-588
inline bool check_sumcheck_product(const Fr &h, const std::vector<Triple> &proof, const FrVec &challenge, size_t rounds) {
    if (proof.size() < rounds || challenge.size() + 1 < rounds) return false;
    if (!rounds) return true;
    if (proof[0][0] + proof[0][1] != h) return false;
    for (size_t i = 1; i < rounds; ++i)
        if (proof[i][0] + proof[i][1] != product_round_target(proof[i - 1], challenge[i - 1])) return false;
    return true;
}

// the first `rounds` rows as a chain with optional pinned ends (see the header comment)
inline bool sumcheck_product_chain(const std::vector<Triple> &proof, const FrVec &challenge, size_t rounds, const Fr *claimed, const Fr *final_eval) {
    if (proof.size() < rounds || challenge.size() < rounds) return false;
    if (!rounds) return true;
    if (claimed && !check_sumcheck_product(*claimed, proof, challenge, 1)) return false;
    if (!check_sumcheck_product(proof[0][0] + proof[0][1], proof, challenge, rounds)) return false;
    return !final_eval || product_round_target(proof[rounds - 1], challenge[rounds - 1]) == *final_eval;
}

// mle.rs:95-103 on a host vector (the leader rounds of d_sumcheck_product fold the parties' last values)
inline Fr fold_host(FrVec v, const FrVec &challenges, size_t from = 0) {
    for (size_t i = from; v.size() > 1; ++i) {
        size_t h = v.size() / 2;
        FrVec nx(h);
        for (size_t j = 0; j < h; ++j) nx[j] = v[j] + challenges.at(i) * (v[j + h] - v[j]);
        v = nx;
    }
    return v.at(0);
}

// ---------------------------------------------------------------------------------------------------------------------
// anchors
// ---------------------------------------------------------------------------------------------------------------------
struct AnchorValues {
    char kind;        // 'p' / 'c' / 'd'
    size_t rounds;    // log2(len): the rounds the device ran
    Fr claimed, f_r, g_r;
    FrVec challenge;  // as traced: `rounds` device challenges (+ log2(N_p) leader rounds for 'd')
};

// (sum_j f_j g_j, f(r), g(r)) through kernels the product sumcheck does not share
inline AnchorValues product_anchor(Ctx &be, const ScTrace &t) {
    size_t n = Ctx::log2_exact(t.len);
    if (t.challenge.size() < n) throw std::invalid_argument("product_anchor: a traced sumcheck of 2^" + std::to_string(n) + " elements carries " + std::to_string(t.challenge.size()) + " challenges");
    FrVec ch(t.challenge.begin(), t.challenge.begin() + n);
    AnchorValues a;
    a.kind = t.kind, a.rounds = n, a.challenge = t.challenge;
    DevPtr prod = be.fr_mul(t.f, t.g, t.len);
    if (!n) {
        a.claimed = be.to_host(prod, 1)[0];
    } else {
        ScResult r = be.sumcheck(prod, t.len, ch);
        a.claimed = r.sums[0] + r.sums[1];
    }
    a.f_r = be.to_host(be.fold(t.f, t.len, ch), 1)[0];
    a.g_r = be.to_host(be.fold(t.g, t.len, ch), 1)[0];
    return a;
}
inline std::vector<AnchorValues> trace_anchor_values(Ctx &be, const std::vector<ScTrace> &trace) {
    std::vector<AnchorValues> out;
    for (auto &t : trace) out.push_back(product_anchor(be, t));
    return out;
}

struct CheckReport {
    size_t transcripts = 0;         // chains checked with both ends pinned
    size_t closing_rows = 0;        // c_sumcheck_product tails recomputed from independently folded values
    size_t recomputed = 0;          // commits / opens compared with a one-call-at-a-time recomputation
    std::vector<std::string> bad;   // labels of everything that failed
    bool ok() const { return bad.empty(); }
};

// What pins the transcripts of one party: its own anchor values, every party's (claimed, f(r), g(r)) of the c / d items, and
// pss2ss of the independently folded last values of every c item.  COLLECTIVE: every party builds it in lock step (one
// all-gather, two pss2ss exchanges per c item).
struct ProductAnchors {
    std::vector<AnchorValues> mine;
    std::vector<FrVec> all;             // [party][3 per c / d item, in trace order]
    std::vector<FrVec> tail_f, tail_g;  // per item: pss2ss(f(r)), pss2ss(g(r)) for 'c', empty otherwise
    size_t n_parties = 1, l = 1;
    bool leader = true;
};
inline ProductAnchors gather_product_anchors(Ctx &be, const std::vector<ScTrace> &trace, const PackedSharingParams &pp, Net &net) {
    ProductAnchors A;
    A.mine = trace_anchor_values(be, trace);
    A.n_parties = net.n_parties, A.l = pp.l, A.leader = net.is_leader();
    FrVec flat;
    for (auto &a : A.mine)
        if (a.kind != 'p') flat.push_back(a.claimed), flat.push_back(a.f_r), flat.push_back(a.g_r);
    A.all = net.all_gather_fr(flat);
    for (auto &a : A.mine) {
        A.tail_f.push_back(a.kind == 'c' ? pss2ss(a.f_r, pp, net) : FrVec{});
        A.tail_g.push_back(a.kind == 'c' ? pss2ss(a.g_r, pp, net) : FrVec{});
    }
    return A;
}

// Every product-sumcheck transcript of a driver's result against its anchored chain.  Trace order == transcript order in every
// driver of hyperplonk.hpp (gate proofs, then wiring proofs).  Pure host arithmetic on the gathered anchors.
//   'c'  rows 0 .. n-1 are this party's share-level rounds (claimed / final = this party's own values); the tail -- log2(l)
//        rounds on pss2ss of the last values and the closing (0, vf vg, 0) row (dsumcheck.rs:224-282) -- is recomputed
//        from pss2ss of the independently folded f(r), g(r) and compared row by row
//   'd'  leader only: rows are sums over the parties plus log2(N_p) leader rounds over the parties' last values (:440-507):
//        claimed = sum_p claimed_p, final = fold(f_p(r)) fold(g_p(r)) over the leader challenges
//   'p'  sumcheck_product (the leader's top-tree sumchecks, dhyperplonk.rs:506-508): closing row (0, f g, 0) = the last target
inline void check_product_transcripts(const ProductAnchors &A, const Transcript &t, CheckReport &rep) {
    std::vector<const std::vector<Triple> *> proofs;
    for (auto &p : t.gate_proofs) proofs.push_back(&p);
    for (auto &p : t.wiring_proofs) proofs.push_back(&p);
    size_t np = A.n_parties, shared = 0, logl = log2_floor(A.l);
    for (size_t e = 0; e < A.mine.size(); ++e) {
        const AnchorValues &a = A.mine[e];
        std::string label = e < t.gate_proofs.size() ? "gate[" + std::to_string(e) + "]" : "wiring[" + std::to_string(e - t.gate_proofs.size()) + "]";
        size_t slot = shared;
        if (a.kind != 'p') ++shared;
        if (e >= proofs.size()) {
            rep.bad.push_back(label + ": no transcript for this traced sumcheck");
            continue;
        }
        const std::vector<Triple> &pr = *proofs[e];
        bool good = true;
        if (a.kind == 'c') {
            Fr fin = a.f_r * a.g_r;
            // (the traced challenge holds the rounds of phase 1; phase 2 re-reads its first log2(l) entries: a table shorter than l -- small n,
            // large l -- has fewer, which is a labelled failure of the check, not an exception out of it)
            good = pr.size() == a.rounds + logl + 1 && a.challenge.size() >= std::max(a.rounds, logl) && sumcheck_product_chain(pr, a.challenge, a.rounds, &a.claimed, &fin);
            if (good) {
                FrVec vf = A.tail_f[e], vg = A.tail_g[e];
                for (size_t i = 0; i < logl && good; ++i) {
                    Triple want = detail::round_product(vf, vg, a.challenge.at(i));  // phase 2 re-uses challenge[0 ..] (:129, :232)
                    good = pr[a.rounds + i] == want;
                }
                const Triple &last = pr.back();
                good = good && last[0].is_zero() && last[2].is_zero() && last[1] == vf[0] * vg[0];
            }
            ++rep.closing_rows;
        } else if (a.kind == 'd') {
            if (!A.leader) {
                if (!pr.empty()) rep.bad.push_back(label + ": a worker holds d_sumcheck_product rows");
                continue;
            }
            size_t s = log2_floor(np);
            Fr claimed = Fr::zero();
            FrVec fs, gs;
            for (size_t p = 0; p < np; ++p) {
                claimed += A.all[p].at(3 * slot);
                fs.push_back(A.all[p].at(3 * slot + 1)), gs.push_back(A.all[p].at(3 * slot + 2));
            }
            good = pr.size() == a.rounds + s && a.challenge.size() >= a.rounds + s;
            if (good) {
                Fr fin = fold_host(fs, a.challenge, a.rounds) * fold_host(gs, a.challenge, a.rounds);
                good = sumcheck_product_chain(pr, a.challenge, a.rounds + s, &claimed, &fin);
            }
        } else {
            Fr fin = a.f_r * a.g_r;
            good = pr.size() == a.rounds + 1 && sumcheck_product_chain(pr, a.challenge, a.rounds, &a.claimed, &fin);
            if (good) good = pr.back()[0].is_zero() && pr.back()[2].is_zero() && pr.back()[1] == fin;
        }
        ++rep.transcripts;
        if (!good) rep.bad.push_back(label);
    }
    // every transcript the party holds must have been traced (workers hold empty d_ transcripts and no top-tree ones)
    size_t held = 0, traced_held = 0;
    for (auto *p : proofs) held += !p->empty();
    for (size_t e = 0; e < A.mine.size() && e < proofs.size(); ++e) traced_held += !proofs[e]->empty();
    if (held != traced_held) rep.bad.push_back("transcripts without an anchor: " + std::to_string(held) + " held, " + std::to_string(traced_held) + " traced");
}

// the transcript shape of dhyperplonk (dhyperplonk.rs:159-571): 6 gate proofs; wiring: 1 (2.c) + 3 (2.e.1) + 3 (n - s)
// layered + 3 leader-tree proofs on the leader, 1 + empty ones on a worker
inline void check_dhyperplonk_shape(size_t n, const Transcript &t, Net &net, CheckReport &rep) {
    size_t s = log2_floor(net.n_parties);
    if (t.gate_proofs.size() != 6) rep.bad.push_back("gate: " + std::to_string(t.gate_proofs.size()) + " proofs instead of 6");
    if (t.gate_commitments.size() != 6) rep.bad.push_back("gate: " + std::to_string(t.gate_commitments.size()) + " (commitment, opening) pairs instead of 6");
    size_t want = 1 + 3 + 3 * (n - s) + (net.is_leader() ? 3 : 0);
    if (t.wiring_proofs.size() != want) rep.bad.push_back("wiring: " + std::to_string(t.wiring_proofs.size()) + " proofs instead of " + std::to_string(want));
}

// sampled commits / opens of a dhyperplonk transcript against the one-call-at-a-time forms of dist_primitive.hpp (other
// batch shapes, other window classes: the batched driver must put every output at the reference's position).  COLLECTIVE.
inline void check_dhyperplonk_recompute(size_t n, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, const Transcript &t, CheckReport &rep) {
    size_t M = size_t(1) << n, hlen = 4 * M / net.n_parties;
    auto same_open = [](const Opening &a, const Opening &b) { return a.value == b.value && a.proofs == b.proofs; };
    auto expect = [&](bool ok, const char *what) {
        ++rep.recomputed;
        if (!ok) rep.bad.push_back(what);
    };
    expect(t.wiring_commits.size() > 0 && t.wiring_commits[0] == d_commit(be, pk.d_commitment, pk.T("local_s_p"), hlen, net), "d_commit(local_s) differs from the single call");
    expect(t.wiring_opens.size() > 4 && same_open(t.wiring_opens[4], d_open(be, pk.d_commitment, pk.T("sid_p"), hlen, pk.challenge_r2, net)), "d_open(sid_p) differs from the single call");
    expect(t.wiring_opens.size() > 1 && same_open(t.wiring_opens[1], c_open(be, pk.c_commitment, pk.T("V"), 4 * M / pp.l, pk.challenge_r2, pp, net)), "c_open(V) differs from the single call");
    expect(t.gate_commitments.size() > 0 && t.gate_commitments[0].first == c_commit(be, pk.c_commitment, {pk.T("a_evals")}, {pk.L("a_evals")}, pp, net)[0], "c_commit(a) differs from the single call");
    expect(t.gate_commitments.size() > 5 && same_open(t.gate_commitments[5].second, d_open(be, pk.d_commitment, pk.T("S2_p"), pk.L("S2_p"), pk.challenge, net)), "d_open(S2_p) differs from the single call");
}

// the same for a cpermcheck transcript (dhyperplonk.rs:1249-1385): shape, the public wires' and num's commitments / openings against
// the one-call-at-a-time forms (the driver runs them inside ONE pass and ONE batch), and the repeated opens of num / den (:1324 and
// :1371: same table, same point -- the driver computes them once) against each other and against a single call.  COLLECTIVE.
inline void check_cpermcheck_recompute(size_t n, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, const Transcript &t, CheckReport &rep) {
    size_t G4 = 4 * ((size_t(1) << n) / pp.l);
    auto same_open = [](const Opening &a, const Opening &b) { return a.value == b.value && a.proofs == b.proofs; };
    auto expect = [&](bool ok, const char *what) {
        ++rep.recomputed;
        if (!ok) rep.bad.push_back(what);
    };
    if (t.wiring_proofs.size() != 6 || t.wiring_commits.size() != 10 || t.wiring_opens.size() != 12) {
        rep.bad.push_back("cpermcheck: expected 6 proofs, 10 commitments, 12 openings");
        return;
    }
    size_t rounds = Ctx::log2_exact(G4) + log2_floor(pp.l);
    for (auto &o : t.wiring_opens)
        if (o.proofs.size() != rounds) rep.bad.push_back("cpermcheck: an opening with the wrong number of proofs");
    expect(t.wiring_commits[0] == c_commit(be, pk.c_commitment, {pk.T("ssigma")}, {G4}, pp, net)[0], "c_commit(ssigma) differs from the single call");
    expect(same_open(t.wiring_opens[1], c_open(be, pk.c_commitment, pk.T("sid"), G4, pk.challenge_r1, pp, net)), "c_open(sid) differs from the single call");
    DevPtr num = be.fr_axpb(pk.T("V"), pk.T("sid"), pk.alpha, pk.beta, G4);  // :1277-1279
    expect(t.wiring_commits[2] == c_commit(be, pk.c_commitment, {num}, {G4}, pp, net)[0], "c_commit(num) differs from the single call");
    expect(same_open(t.wiring_opens[6], c_open(be, pk.c_commitment, num, G4, pk.challenge_r1, pp, net)), "the second c_open(num) differs from the single call");
    expect(same_open(t.wiring_opens[2], t.wiring_opens[6]) && same_open(t.wiring_opens[7], t.wiring_opens[11]), "the two opens of num / den differ");
}

}  // namespace zkhost
