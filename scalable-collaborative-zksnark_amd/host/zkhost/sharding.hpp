// Strong-scaling ONE primitive across G GPUs from the C++ host: one party object per GPU; every
// function has exactly one exchange step (an all-gather of a few hundred bytes) because the partial results combine linearly.
//
//   sharded_msm                 contiguous chunks of (base, scalar) pairs per rank (the SRS chunk stays resident); the G partial
//                               points are all-gathered and added
//   sharded_sumcheck(_product)  CYCLIC layout: global index i lives on rank i mod G at local slot i div G.  Pairs (j, j + m/2)
//                               stay on one rank while m/2 >= G, so each rank runs the unmodified kernel for log2(N / G) rounds
//                               with challenge[0..]; the per-round sums add up across ranks, and the G leftovers (rank r holding
//                               global index r) finish the last log2(G) rounds.  The transcript is bit-identical to the monolithic
//                               sumcheck(_product) on the full table (dsumcheck.rs:6-26, 28-90) -- unlike d_sumcheck, whose
//                               contiguous chunks consume the variables in a different order.
#pragma once
#include "dist_primitive.hpp"

namespace zkhost {

// sum over ranks of MSM(bases chunk, scalars chunk), on every rank
inline G1 sharded_msm(Ctx &be, const Srs &srs_chunk, const DevPtr &scalars_chunk, size_t n_local, Net &net) {
    std::vector<G1Vec> got = net.all_gather_g1(G1Vec{be.msm_g1(srs_chunk, scalars_chunk, n_local)});
    return be.g1_lincomb_batch(detail::by_item(got), FrVec(net.n_parties, Fr::raw_u64(1)), 1)[0];
}

// the slice of a full table owned by `rank` under the cyclic layout
inline FrVec cyclic_shard(const FrVec &table, size_t rank, size_t world) {
    FrVec out;
    for (size_t i = rank; i < table.size(); i += world) out.push_back(table[i]);
    return out;
}

// == sumcheck(full table, challenge) on every rank: n + 1 pairs
inline std::vector<Pair> sharded_sumcheck(Ctx &be, const DevPtr &local_tab, size_t local_len, const FrVec &challenge, Net &net) {
    size_t nl = Ctx::log2_exact(local_len), g = log2_floor(net.n_parties);
    ScResult r = be.sumcheck(local_tab, local_len, challenge);
    FrVec mine = r.sums;
    mine.push_back(r.last_f);
    std::vector<FrVec> all = net.all_gather_fr(mine);
    std::vector<Pair> out(nl, Pair{Fr::zero(), Fr::zero()});
    FrVec v;
    for (auto &a : all) {
        for (size_t i = 0; i < nl; ++i) out[i][0] += a[2 * i], out[i][1] += a[2 * i + 1];
        v.push_back(a[2 * nl]);
    }
    for (size_t i = nl; i < nl + g; ++i) out.push_back(detail::round_plain(v, challenge.at(i)));
    out.push_back({Fr::zero(), v[0]});
    return out;
}

// == sumcheck_product(full f, full g, challenge) on every rank: n + 1 triples
inline std::vector<Triple> sharded_sumcheck_product(Ctx &be, const DevPtr &local_f, const DevPtr &local_g, size_t local_len, const FrVec &challenge, Net &net) {
    size_t nl = Ctx::log2_exact(local_len), g = log2_floor(net.n_parties);
    ScResult r = be.sumcheck_product(local_f, local_g, local_len, challenge);
    FrVec mine = r.sums;
    mine.push_back(r.last_f), mine.push_back(r.last_g);
    std::vector<FrVec> all = net.all_gather_fr(mine);
    std::vector<Triple> out(nl, Triple{Fr::zero(), Fr::zero(), Fr::zero()});
    FrVec f, gg;
    for (auto &a : all) {
        for (size_t i = 0; i < nl; ++i)
            for (size_t c = 0; c < 3; ++c) out[i][c] += a[3 * i + c];
        f.push_back(a[3 * nl]), gg.push_back(a[3 * nl + 1]);
    }
    for (size_t i = nl; i < nl + g; ++i) out.push_back(detail::round_product(f, gg, challenge.at(i)));
    out.push_back({Fr::zero(), f[0] * gg[0], Fr::zero()});
    return out;
}

}  // namespace zkhost
