// Test driver of the C++ host mirror (scalable-collaborative-zksnark_amd/host/zkhost): runs every `dist-primitive` function
// of the mirror on inputs written by pytest and writes the outputs back; tests/test_host_cpp.py runs the same sequence
// through the Python host layer (and, on the CPU, the oracle) and compares the two record files bit for bit.
//
//   host_mirror host <in> <out>     no GPU: field arithmetic, PackedSharingParams, the leader rounds, merge / transpose / sub_index
//   host_mirror gpu  <in> <out>     all parties as threads of this process, one ctx each on GPU 0 (LocalTestNet), or party 0 on
//                                   the leader-echo net; every collaborative primitive once
//   host_mirror proof <in> <out>    the protocol drivers (zkhost/hyperplonk.hpp) on tables written by pytest: params = l, n, echo,
//                                   which (0 dhyperplonk, 1 data-parallel, 2 dpermcheck, 3 cpermcheck); the transcript of every party
// Record file: { char name[24]; u64 party; u64 nbytes; payload } ...
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "zkhost/hyperplonk.hpp"
#include "zkhost/serialize.hpp"
#include "zkhost/sharding.hpp"

using namespace zkhost;

struct Records {
    std::map<std::pair<std::string, uint64_t>, Bytes> rec;
    std::vector<std::pair<std::string, uint64_t>> order;
    std::mutex m;

    void put(const std::string &name, uint64_t party, const void *data, size_t bytes) {
        std::lock_guard<std::mutex> lk(m);
        auto key = std::make_pair(name, party);
        if (!rec.count(key)) order.push_back(key);
        Bytes &b = rec[key];
        b.insert(b.end(), (const uint8_t *)data, (const uint8_t *)data + bytes);
    }
    void put(const std::string &n, uint64_t p, const FrVec &v) { put(n, p, v.data(), 32 * v.size()); }
    void put(const std::string &n, uint64_t p, const Fr &v) { put(n, p, v.v, 32); }
    void put(const std::string &n, uint64_t p, const G1Vec &v) { put(n, p, v.data(), 144 * v.size()); }
    void put(const std::string &n, uint64_t p, const G1 &v) { put(n, p, v.data(), 144); }
    void put(const std::string &n, uint64_t p, const std::vector<Pair> &v) { put(n, p, v.data(), 64 * v.size()); }
    void put(const std::string &n, uint64_t p, const std::vector<Triple> &v) { put(n, p, v.data(), 96 * v.size()); }
    void put(const std::string &n, uint64_t p, const Opening &o) {
        put(n, p, o.value);
        put(n, p, o.proofs);
    }
    void put_u64(const std::string &n, uint64_t p, uint64_t x) { put(n, p, &x, 8); }

    bool load(const char *path) {
        FILE *f = std::fopen(path, "rb");
        if (!f) return false;
        char name[24];
        uint64_t hdr[2];
        while (std::fread(name, 1, 24, f) == 24 && std::fread(hdr, 8, 2, f) == 2) {
            Bytes b(hdr[1]);
            if (hdr[1] && std::fread(b.data(), 1, hdr[1], f) != hdr[1]) break;
            name[23] = 0;
            rec[{name, hdr[0]}] = b;
        }
        std::fclose(f);
        return true;
    }
    bool save(const char *path) {
        FILE *f = std::fopen(path, "wb");
        if (!f) return false;
        for (auto &key : order) {
            char name[24] = {0};
            std::strncpy(name, key.first.c_str(), 23);
            const Bytes &b = rec[key];
            uint64_t hdr[2] = {key.second, b.size()};
            std::fwrite(name, 1, 24, f);
            std::fwrite(hdr, 8, 2, f);
            if (!b.empty()) std::fwrite(b.data(), 1, b.size(), f);
        }
        std::fclose(f);
        return true;
    }
    FrVec fr(const std::string &name, uint64_t party = 0) const {
        auto it = rec.find({name, party});
        if (it == rec.end()) throw std::runtime_error("input record missing: " + name);
        FrVec v(it->second.size() / 32);
        std::memcpy(v.data(), it->second.data(), 32 * v.size());
        return v;
    }
    uint64_t u64(const std::string &name, size_t i = 0) const {
        auto it = rec.find({name, 0});
        if (it == rec.end()) throw std::runtime_error("input record missing: " + name);
        uint64_t x;
        std::memcpy(&x, it->second.data() + 8 * i, 8);
        return x;
    }
};

static FrVec flatten(const std::vector<FrVec> &m) {
    FrVec f;
    for (auto &r : m) f.insert(f.end(), r.begin(), r.end());
    return f;
}

// ---- no GPU: everything of the mirror that is plain host code ----
static void run_host(const Records &in, Records &out) {
    FrVec a = in.fr("a"), b = in.fr("b");
    FrVec sum, dif, prod, inv, canon;
    for (size_t i = 0; i < a.size(); ++i) {
        sum.push_back(a[i] + b[i]);
        dif.push_back(a[i] - b[i]);
        prod.push_back(a[i] * b[i]);
        inv.push_back(a[i].is_zero() ? Fr::zero() : a[i].inverse());
        canon.push_back(a[i].to_canonical());
    }
    out.put("add", 0, sum), out.put("sub", 0, dif), out.put("mul", 0, prod), out.put("inv", 0, inv), out.put("canonical", 0, canon);
    out.put("root_of_unity", 0, fr_two_adic_root());
    out.put("splitmix_fr", 0, SplitMix64(0x5CA1AB1Eull + 100001).fr_vec(300));  // the synthetic-table generator
    size_t nl = in.rec.at({"ls", 0}).size() / 8;
    for (size_t i = 0; i < nl; ++i) {
        size_t l = in.u64("ls", i);
        PackedSharingParams pp(l);
        out.put("pack_matrix", l, flatten(pp.pack_matrix));
        out.put("unpack_matrix", l, flatten(pp.unpack_matrix));
        out.put("unpack2_matrix", l, flatten(pp.unpack2_matrix));
        FrVec secrets(a.begin(), a.begin() + l), shares(b.begin(), b.begin() + pp.n);
        out.put("pack_from_public", l, pp.pack_from_public(secrets));
        out.put("pack_single", l, pp.pack_single(a[0]));
        out.put("unpack", l, pp.unpack(shares));
        out.put("unpack2", l, pp.unpack2(shares));
        out.put("dmsm_coeffs", l, pp.dmsm_coeffs(pp.n - 1));
        out.put("degree_reduce_row", l, pp.degree_reduce_row(1));
        for (auto kind : {PackedSharingParams::Map::Pack, PackedSharingParams::Map::Unpack, PackedSharingParams::Map::Unpack2}) {
            NttTables t = pp.ntt_tables(kind);
            out.put("ntt_winv", l, t.winv), out.put("ntt_w", l, t.w), out.put("ntt_scale", l, t.scale);
        }
        // all parties as threads: pss2ss, degree_reduce and the d_unpack family need no device
        std::vector<FrVec> r_pss(pp.n), r_un(pp.n), r_un2(pp.n);
        FrVec r_dr(pp.n), r_u0(pp.n);
        std::vector<std::array<uint64_t, 2>> comm(pp.n);
        LocalTestNet::simulate_network_round(pp.n, [&](size_t p, LocalTestNet &net) {
            r_pss[p] = pss2ss(b[p], pp, net);
            r_dr[p] = degree_reduce(b[p], pp, net);
            r_u0[p] = d_unpack_0(b[p], pp, net);
            r_un[p] = d_unpack(b[p], 2, pp, net);
            r_un2[p] = d_unpack2(b[p], 3, pp, net);
            comm[p] = {net.upload, net.download};
        });
        out.put("pss2ss", l, flatten(r_pss)), out.put("degree_reduce", l, r_dr), out.put("d_unpack_0", l, r_u0);
        out.put("d_unpack", l, flatten(r_un)), out.put("d_unpack2", l, flatten(r_un2));
        out.put("comm", l, comm.data(), 16 * comm.size());
        LeaderEchoNet echo(pp.n);
        out.put("pss2ss_echo", l, pss2ss(b[0], pp, echo));
    }
    // the leader rounds of the collaborative sumchecks on host vectors
    FrVec f(a.begin(), a.begin() + 8), g(b.begin(), b.begin() + 8);
    for (size_t i = 0; i < 3; ++i) {
        Triple t = detail::round_product(f, g, a[8 + i]);
        out.put("round_product", 0, t.data(), 96);
    }
    FrVec v(a.begin(), a.begin() + 8);
    for (size_t i = 0; i < 3; ++i) {
        Pair s = detail::round_plain(v, b[8 + i]);
        out.put("round_plain", 0, s.data(), 64);
    }
    out.put("round_last", 0, FrVec{f[0], g[0], v[0]});
    // wire formats (zkhost/serialize.hpp): Fr, Vec<Fr>, G1 compressed / uncompressed (+ round trips), the delegator's share files
    {
        out.put("fr_vec_serialize", 0, fr_vec_serialize(a).data(), 8 + 32 * a.size());
        if (fr_vec_deserialize(fr_vec_serialize(a)) != a) throw std::runtime_error("Vec<Fr> round trip");
        const Bytes &raw = in.rec.at({"points", 0});
        G1Vec pts(raw.size() / 144);
        std::memcpy(pts.data(), raw.data(), raw.size());
        Bytes comp, unc;
        for (const G1 &p : pts) {
            G1Affine af = g1_affine(p);
            size_t c0 = comp.size(), u0 = unc.size();
            g1_serialize_compressed(af, comp), g1_serialize_uncompressed(af, unc);
            G1Affine b1 = g1_deserialize_compressed(&comp[c0]), b2 = g1_deserialize_uncompressed(&unc[u0]);
            for (const G1Affine &b : {b1, b2})
                if (b.infinity != af.infinity || (!af.infinity && (b.x != af.x || b.y != af.y))) throw std::runtime_error("G1 round trip");
        }
        out.put("g1_compressed", 0, comp.data(), comp.size()), out.put("g1_uncompressed", 0, unc.data(), unc.size());
        Bytes vec = g1_vec_serialize_compressed(pts);
        out.put("g1_vec_compressed", 0, vec.data(), vec.size());
        if (in.rec.count({"dir", 0})) {
            const Bytes &d = in.rec.at({"dir", 0});
            delegator_write(std::string(d.begin(), d.end()), a, PackedSharingParams(2));
        }
    }
    // merge (dacc_product.rs:416-428) of 3 vectors of 7, sub_index (:18-23; KAT :442-448), transpose (operator.rs:42-49)
    std::vector<FrVec> parts;
    for (size_t q = 0; q < 3; ++q) parts.emplace_back(a.begin() + 7 * q, a.begin() + 7 * q + 7);
    out.put("merge", 0, merge(parts));
    for (size_t i = 1; i < 16; ++i) {
        auto ab = sub_index(i);
        out.put_u64("sub_index", 0, ab.first), out.put_u64("sub_index", 0, ab.second);
    }
    std::vector<std::vector<uint64_t>> m = {{1, 2, 3}, {4, 5, 6}};
    for (auto &row : transpose(m))
        for (uint64_t x : row) out.put_u64("transpose", 0, x);
}

// ---- one party's pass over every collaborative primitive ----
static void run_party(const Records &in, Records &out, size_t p, Net &net, const PackedSharingParams &pp) {
    size_t m = in.u64("params", 1), M = size_t(1) << m, P = pp.n, logP = log2_floor(P), logl = log2_floor(pp.l);
    Ctx be(0);
    FrVec chal = in.fr("chal"), point = in.fr("point");
    DevPtr f = be.to_device(in.fr("f", p)), g = be.to_device(in.fr("g", p));
    DevPtr h0 = be.to_device(in.fr("h0", p)), h1 = be.to_device(in.fr("h1", p)), h2 = be.to_device(in.fr("h2", p));
    PolynomialCommitmentCub pc_d = PolynomialCommitmentCub::new_random(be, m + logP, P, 3);
    PolynomialCommitmentCub pc_c = PolynomialCommitmentCub::new_single(be, m + logl, pp, 5);
    const PowersOfG &gd = pc_d.mature(), &gc = pc_c.mature();
    Fr share = in.fr("f", p)[0];

    out.put("sumcheck", p, sumcheck(be, f, M, chal));
    out.put("sumcheck_product", p, sumcheck_product(be, f, g, M, chal));
    out.put("pss2ss", p, pss2ss(share, pp, net));
    out.put("c_sumcheck", p, c_sumcheck(be, f, M, chal, pp, net));
    out.put("c_sumcheck_product", p, c_sumcheck_product(be, f, g, M, chal, pp, net));
    out.put("d_sumcheck", p, d_sumcheck(be, f, M, chal, net));
    out.put("d_sumcheck_product", p, d_sumcheck_product(be, f, g, M, chal, net));

    std::vector<SrsPtr> bases = {gc[m + logl], gc[m - 1 + logl]};
    out.put("d_msm", p, d_msm(be, bases, {f, g}, {M, M / 2}, pp, net));
    out.put("d_msm_unscaled", p, d_msm(be, bases, {f, g}, {M, M / 2}, pp, net, false));
    out.put("commit", p, commit(be, gd, f, M));
    out.put("open", p, open(be, gd, f, M, point));
    out.put("d_commit", p, d_commit(be, gd, f, M, net));
    out.put("d_open", p, d_open(be, gd, f, M, point, net));
    out.put("c_commit", p, c_commit(be, gc, {f, g.fr(M / 2)}, {M, M / 2}, pp, net));
    out.put("c_open", p, c_open(be, gc, f, M, point, pp, net));

    ProductTree tree = acc_product(be, f, M);
    out.put("acc_product", p, be.to_host(tree.tree, 2 * M));
    auto views = tree.views(be);
    for (auto &v : views) out.put("acc_product_views", p, be.to_host(v, M));
    auto dap = d_acc_product(be, g, M, net);
    out.put("d_acc_product", p, be.to_host(dap.first.tree, 2 * M));
    out.put("d_acc_product_top", p, dap.second ? *dap.second : FrVec{});
    auto cap = c_acc_product(be, g, M, pp, net);
    out.put("c_acc_product", p, be.to_host(cap.first.tree, 2 * M));
    out.put("c_acc_product_top", p, cap.second ? *cap.second : FrVec{});

    out.put("fix_variable", p, be.to_host(fix_variable(be, f, M, FrVec(point.begin(), point.begin() + 3)), M >> 3));
    FixedVariable fv = d_fix_variable(be, f, M, FrVec(point.begin(), point.begin() + m + logl), pp, net);
    out.put("d_fix_variable", p, fv.value ? *fv.value : be.to_host(fv.table, fv.len)[0]);
    FixedVariable fv2 = d_fix_variable(be, f, M, FrVec(point.begin(), point.begin() + 2), pp, net);
    out.put("d_fix_variable_short", p, be.to_host(fv2.table, fv2.len));

    FrVec g_host = in.fr("g", p), few(g_host.begin(), g_host.begin() + 5);
    out.put("degree_reduce", p, degree_reduce(share, pp, net));
    out.put("degree_reduce_many", p, degree_reduce_many(be, few, pp, net));
    out.put("d_unpack_0", p, d_unpack_0(share, pp, net));
    out.put("d_unpack", p, d_unpack(share, P - 1, pp, net));
    out.put("d_unpack2", p, d_unpack2(share, 1 % P, pp, net));
    out.put("d_unpack2_many", p, d_unpack2_many(be, few, 0, pp, net));

    // the structured parameter set (PolynomialCommitmentCub::new, dpoly_comm.rs:37-67) and this party's packed share of it (:164-194)
    {
        PolynomialCommitmentCub cub = PolynomialCommitmentCub::make(be, FrVec(chal.begin(), chal.begin() + m + logl));
        PolynomialCommitmentCub packed = cub.to_packed(be, pp, p);
        out.put("structured_commit", p, commit(be, cub.mature(), f, M));
        out.put("structured_c_commit", p, c_commit(be, packed.mature(), {f}, {M}, pp, net));
        out.put("structured_c_open", p, c_open(be, packed.mature(), f, M, point, pp, net));
    }
    // strong-scaling shards (zkhost/sharding.hpp): the parties' tables are the cyclic shards of one table of P M elements
    out.put("sharded_msm", p, sharded_msm(be, *gd[m], f, M, net));
    out.put("shard_sc", p, sharded_sumcheck(be, f, M, chal, net));
    out.put("shard_sc_product", p, sharded_sumcheck_product(be, f, g, M, chal, net));
    if (p == 0 && !net.echo) {  // ... and equal the monolithic calls on that table
        FrVec ff(P * M), fg(P * M);
        for (size_t q = 0; q < P; ++q) {
            FrVec tf = in.fr("f", q), tg = in.fr("g", q);
            for (size_t i = 0; i < M; ++i) ff[q + P * i] = tf[i], fg[q + P * i] = tg[i];
        }
        if (cyclic_shard(ff, 1 % P, P) != in.fr("f", 1 % P)) throw std::runtime_error("cyclic_shard");
        DevPtr dff = be.to_device(ff), dfg = be.to_device(fg);
        out.put("mono_sc", 0, sumcheck(be, dff, P * M, chal));
        out.put("mono_sc_product", 0, sumcheck_product(be, dff, dfg, P * M, chal));
    }
    if (p == 0 && in.rec.count({"dir", 0})) {  // a share file onto the device and back (Montgomery conversion on the GPU)
        const Bytes &d = in.rec.at({"dir", 0});
        std::string dir(d.begin(), d.end());
        auto [buf, cnt] = fr_file_to_device(be, dir + "/worker_3");
        out.put("share_file_on_device", 0, be.to_host(buf, cnt));
        fr_device_to_file(be, buf, cnt, dir + "/worker_3.copy");
    }
    auto sh = c_acc_product_and_share(be, f, g, h0, h1, h2, M, pp, net);
    for (auto &s : sh) {
        out.put_u64("c_acc_share_len", p, s.len);
        out.put("c_acc_product_and_share", p, be.to_host(s.buf, s.len));
    }
    out.put_u64("comm", p, net.upload), out.put_u64("comm", p, net.download);
}

// ---- one party's run of a protocol driver ----
static void put_transcript(Records &out, size_t p, const Transcript &t) {
    for (auto &pr : t.gate_proofs) out.put("gate_proofs", p, pr);
    for (auto &co : t.gate_commitments) out.put("gate_commitments", p, co.first), out.put("gate_commitments", p, co.second);
    for (auto &pr : t.wiring_proofs) out.put("wiring_proofs", p, pr);
    out.put("wiring_commits", p, t.wiring_commits);
    for (auto &o : t.wiring_opens) out.put("wiring_opens", p, o);
    for (const char *name : {"gate_proofs", "gate_commitments", "wiring_proofs", "wiring_commits", "wiring_opens"}) out.put(name, p, nullptr, 0);  // (empty lists still get a record)
}

static void run_proof(const Records &in, Records &out, size_t p, Net &net, const PackedSharingParams &pp) {
    size_t n = in.u64("params", 1), which = in.u64("params", 3);
    Ctx be(0);
    PackedProvingParameters pk;
    pk.n = n;
    for (auto &nl : PackedProvingParameters::layout(n, pp)) {
        FrVec v = in.fr(nl.first, p);
        if (v.size() != nl.second) throw std::runtime_error("table " + nl.first + " has an unexpected length");
        pk.put(be, nl.first, v);
    }
    if (which == 3)
        for (const char *name : {"mask", "unmask0", "unmask1", "unmask2"}) pk.put(be, name, in.fr(name, p));
    pk.set_challenges(in.fr("chal"));
    pk.finish_setup(be, pp, in.u64("seeds", p));
    Timers tm;
    Transcript t = which == 3 ? cpermcheck(n, pk, pp, be, net, &tm) : which == 2 ? dpermcheck(n, pk, pp, be, net, &tm) : dhyperplonk(n, pk, pp, be, net, &tm, which == 1);
    put_transcript(out, p, t);
    out.put_u64("comm", p, net.upload), out.put_u64("comm", p, net.download);
    if (p == 0)
        for (auto &kv : tm.t) std::printf("timer %-24s %.6f s\n", kv.first.c_str(), kv.second);
}

// ---- RcclNet on a world of ONE party (all a one-GPU box can form): the C-ABI communicator under the C++ net, and d_msm as the
// single zk_d_msm call it becomes when the communicator lives in the compute ctx -- checked against the queue-free local form ----
static int run_rccl1() {
    Ctx be(0);
    uint8_t id[ZK_COMM_ID_BYTES];
    be.check(zk_comm_unique_id(id));
    RcclNet net(be, 0, 1, id);
    PackedSharingParams pp(1);
    size_t M = 1 << 10;
    FrVec f = SplitMix64(77).fr_vec(M), g = SplitMix64(78).fr_vec(M / 2);
    DevPtr df = be.to_device(f), dg = be.to_device(g);
    PowersOfG pg = PolynomialCommitmentCub::new_single(be, 10, pp, 9).mature();
    G1Vec got = d_msm(be, {pg[10], pg[9]}, {df, dg}, {M, M / 2}, pp, net);  // -> zk_d_msm (lambda_0 on the device, c_0 on the sum)
    G1Vec local = be.msm_g1_batch({pg[10].get(), pg[9].get()}, {df, dg}, {M, M / 2});
    G1Vec want = be.g1_lincomb_batch(local, FrVec{(pp.c(0) * pp.lambda(0)).to_canonical()}, 2);
    if (got != want) return std::fprintf(stderr, "rccl1: d_msm over RcclNet differs from c_0 lambda_0 MSM\n"), 1;
    // G::msm on host slices (dmsm.rs:23): the generator 5 times with scalars 1..5 -> 15 G; a length mismatch -> Err(min_len)
    {
        G1Affine gen = g1_deserialize_compressed((const uint8_t *)"\x97\xf1\xd3\xa7\x31\x97\xd7\x94\x26\x95\x63\x8c\x4f\xa9\xac\x0f\xc3\x68\x8c\x4f\x97\x74\xb9\x05\xa1\x4e\x3a\x3f\x17\x1b\xac\x58\x6c\x55\xe8\x3f\xf9\x7a\x1a\xef\xfb\x3a\xf0\x0a\xdb\x22\xc6\xbb");
        uint8_t recs[5 * 96];
        FrVec sc;
        for (int i = 0; i < 5; ++i) g1_affine_record(gen, recs + 96 * i), sc.push_back(Fr::from_u64(i + 1));
        G1 s15 = be.msm_g1_host(recs, 96, 5, sc);
        G1 one = be.msm_g1_host(recs, 96, 1, FrVec{Fr::from_u64(1)});
        if (be.g1_lincomb_batch(G1Vec{one}, FrVec{Fr::raw_u64(15)}, 1)[0] != s15) return std::fprintf(stderr, "rccl1: msm_g1_host != 15 G\n"), 1;
        try {
            be.msm_g1_host(recs, 96, 5, FrVec(3, Fr::one()));
            return std::fprintf(stderr, "rccl1: a length mismatch must throw\n"), 1;
        } catch (const MsmLengthError &e) {
            if (e.min_len != 3) return std::fprintf(stderr, "rccl1: Err(min_len) = %zu, expected 3\n", e.min_len), 1;
        }
    }
    // the exchanges: a world of one returns its own data, through HBM
    std::vector<FrVec> ag = net.all_gather_fr(FrVec(f.begin(), f.begin() + 5));
    DevPtr a2a = net.all_to_all_device(be, df, 32 * 7);
    if (ag.size() != 1 || ag[0] != FrVec(f.begin(), f.begin() + 5) || be.to_host(a2a, 7) != FrVec(f.begin(), f.begin() + 7)) return std::fprintf(stderr, "rccl1: exchange mismatch\n"), 1;
    if (net.upload != 0 || net.download != 0) return std::fprintf(stderr, "rccl1: a world of one moves no bytes\n"), 1;
    std::printf("host_mirror ok: rccl world of one\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 2 && !std::strcmp(argv[1], "rccl1")) {
        try {
            return zk_device_count() > 0 ? run_rccl1() : (std::fprintf(stderr, "host_mirror: no GPU visible -- no CPU fallback\n"), 2);
        } catch (const std::exception &e) {
            return std::fprintf(stderr, "host_mirror: %s\n", e.what()), 1;
        }
    }
    if (argc != 4) {
        std::fprintf(stderr, "usage: host_mirror host|gpu|proof <in> <out>\n");
        return 64;
    }
    Records in, out;
    if (!in.load(argv[2])) {
        std::fprintf(stderr, "host_mirror: cannot read %s\n", argv[2]);
        return 66;
    }
    try {
        if (!std::strcmp(argv[1], "host")) {
            run_host(in, out);
        } else {
            if (zk_device_count() <= 0) {
                std::fprintf(stderr, "host_mirror: no GPU visible -- the host mirror has no CPU fallback (zk_device_count = %d)\n", zk_device_count());
                return 2;
            }
            PackedSharingParams pp(in.u64("params", 0));
            auto run = !std::strcmp(argv[1], "proof") ? run_proof : run_party;
            if (in.u64("params", 2)) {  // the no-`comm` fake: party 0 alone
                LeaderEchoNet net(pp.n);
                run(in, out, 0, net, pp);
            } else {
                LocalTestNet::simulate_network_round(pp.n, [&](size_t p, LocalTestNet &net) { run(in, out, p, net, pp); });
            }
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_mirror: %s\n", e.what());
        return 1;
    }
    if (!out.save(argv[3])) return 73;
    std::printf("host_mirror ok: %zu records\n", out.order.size());
    return 0;
}
