// Collaborative HyperPlonk call sequence -- C++ host mirror of hyperplonk/src/dhyperplonk.rs (`PackedProvingParameters::new`
// :65-156, `dhyperplonk` :159-571, `dhyperplonk_data_parallel` :573-960, `dpermcheck` :962-1247, `cpermcheck` :1249-1385).
// Like the reference it is a FIXED sequence of dist-primitive calls on synthetic (random) tables: there is no circuit and no
// Fiat-Shamir, every challenge is pre-sampled.  All tables and SRS levels are resident in HBM; every primitive runs through
// libzkhip.so.  The MSM pass of a step is started asynchronously and collected one step later (pipeline.hpp): the outputs
// keep the reference's positions, the timers keep its labels (only the total is comparable once steps overlap).
#pragma once
#include <chrono>
#include <cstdlib>
#include <map>
#include <string>

#include "pipeline.hpp"

namespace zkhost {

// SplitMix64 -> uniform Fr limbs by rejection (SURVEY.md 8(d) "Synthetic inputs"): any canonical limb pattern is the Montgomery
// form of a uniform element, which is what `random_evaluations` (dist-primitive/src/lib.rs:13-18) produces
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    Fr fr() {
        for (;;) {
            Fr x{{next(), next(), next(), next() & 0x7fffffffffffffffull}};
            if (!Fr::geq_mod(x.v)) return x;
        }
    }
    FrVec fr_vec(size_t n) {
        FrVec v(n);
        for (auto &x : v) x = fr();
        return v;
    }
};

class Timers {  // wall-clock sections with the reference's labels (mpc-net/src/utils/timer.rs)
  public:
    std::map<std::string, double> t;
    // diagnostics (`hyperplonk --marks`): host time stamps of the calls inside a proof, seconds since the first mark
    bool keep_marks = false;
    std::vector<std::pair<std::string, double>> marks;
    void mark(const char *what) {
        if (!keep_marks) return;
        auto now = std::chrono::steady_clock::now();
        if (marks.empty()) t0_ = now;
        marks.push_back({what, std::chrono::duration<double>(now - t0_).count()});
    }
    void start(const std::string &label) { stack_.push_back({label, std::chrono::steady_clock::now()}); }
    void end() {
        auto [label, t0] = stack_.back();
        stack_.pop_back();
        t[label] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

  private:
    std::vector<std::pair<std::string, std::chrono::steady_clock::time_point>> stack_;
    std::chrono::steady_clock::time_point t0_;
};

// hyperplonk/src/dhyperplonk.rs:19-63; every table is a device buffer of Fr
struct PackedProvingParameters {
    size_t n = 0;
    std::map<std::string, DevPtr> tables;
    std::map<std::string, size_t> lens;
    FrVec challenge, challenge_r1, challenge_r2;
    Fr alpha, beta, gamma;
    PowersOfG c_commitment;  // levels 0 .. n+2 (new_single, dpoly_comm.rs:197-219)
    PowersOfG d_commitment;  // levels 0 .. n - log2(N_p) + 2 (new_random, dpoly_comm.rs:220-233)

    const DevPtr &T(const std::string &name) const {
        auto it = tables.find(name);
        if (it == tables.end()) throw ZkError(ZK_ERR_INVALID, "PackedProvingParameters: no table " + name);
        return it->second;
    }
    size_t L(const std::string &name) const { return lens.at(name); }
    void put(Ctx &be, const std::string &name, const FrVec &v) {
        tables[name] = be.to_device(v);
        lens[name] = v.size();
    }

    // the table names and lengths of dhyperplonk.rs:65-156 (a_evals, b_evals, c_evals are folds of V, :71-73)
    static std::vector<std::pair<std::string, size_t>> layout(size_t n, const PackedSharingParams &pp) {
        size_t M = size_t(1) << n, l = pp.l, np = pp.n;
        return {{"V", 4 * M / l},        {"I", M / l},           {"I_p", M / np},        {"S1", M / l},          {"S2", M / l},         {"S1_p", M / np},
                {"S2_p", M / np},        {"ssigma", 4 * M / l},  {"ssigma_p", 4 * M / np}, {"sid", 4 * M / l},   {"sid_p", 4 * M / np}, {"eq", M / l},
                {"eq_top_p", 2 * np},    {"eq_r1", 4 * M / l},   {"eq_r1_p", 4 * M / np}, {"eq_r2", 4 * M / l},  {"eq_r2_p", 4 * M / np},
                // "Jump from sky" (:187-190) and the data-parallel s (:603): per-run random data, kept with the tables here
                {"local_s_p", 4 * M / np}, {"local_s_l", 4 * M / np / l}, {"eq_top", np}, {"s_data_parallel", 4 * M / l}};
    }
    // challenge vector of 3n + 7 Fr: challenge (n), challenge_r1 (n + 2), challenge_r2 (n + 2), alpha, beta, gamma
    void set_challenges(const FrVec &ch) {
        if (ch.size() != 3 * n + 7) throw ZkError(ZK_ERR_INVALID, "PackedProvingParameters: 3n + 7 challenge values expected");
        challenge.assign(ch.begin(), ch.begin() + n);
        challenge_r1.assign(ch.begin() + n, ch.begin() + 2 * n + 2);
        challenge_r2.assign(ch.begin() + 2 * n + 2, ch.begin() + 3 * n + 4);
        alpha = ch[3 * n + 4], beta = ch[3 * n + 5], gamma = ch[3 * n + 6];
    }
    // the folds of V and the synthetic SRS (random points in the reference as well); window_tables: build the MSM window
    // table of every level up to 2^table_max_log2 points (setup work like generating the level; results are bit-identical
    // with and without); a level is skipped when its table would leave less than half of the device free (the rule is below)
    // window_bits(len): width of a level's table (0 = the library's pick for a single MSM of that length)
    void finish_setup(Ctx &be, const PackedSharingParams &pp, uint64_t seed, bool window_tables = true, size_t table_max_log2 = 25,
                      const std::function<int(size_t)> &window_bits = nullptr) {
        size_t M = size_t(1) << n, l = pp.l;
        Fr zero = Fr::zero(), one = Fr::one();
        const std::pair<const char *, std::array<Fr, 2>> folds[3] = {{"a_evals", {zero, zero}}, {"b_evals", {zero, one}}, {"c_evals", {one, zero}}};
        for (auto &f : folds) {
            tables[f.first] = be.fold(T("V"), 4 * M / l, FrVec{f.second[0], f.second[1]});  // fix_variable(&V, ..) :71-73
            lens[f.first] = M / l;
        }
        for (size_t i = 0; i < n + 3; ++i) c_commitment.push_back(be.srs_generate(seed * 7919 + 2 * i + 1, seed * 104729 + 2 * i + 3, std::max<size_t>(1, (size_t(1) << i) / l)));
        for (size_t i = 0; i + log2_floor(pp.n) < n + 3; ++i) d_commitment.push_back(be.srs_generate(seed * 6007 + 2 * i + 5, seed * 15485863 + 2 * i + 7, size_t(1) << i));
        if (!window_tables) return;
        std::vector<SrsPtr> all = c_commitment;
        all.insert(all.end(), d_commitment.begin(), d_commitment.end());
        std::stable_sort(all.begin(), all.end(), [](const SrsPtr &a, const SrsPtr &b) { return a->len() < b->len(); });
        // Record layout per level (round 6; the Python host applies the same rule, zkhip/hyperplonk.py `_synthetic_srs`).  A G1 table record is
        // 96 B packed or 128 B -- one per cache line: the accumulation's gathers move 1 line instead of 1.67, k_accum_tiles -7 .. -10 %
        // (profiles/r05zb_table_rec_ab.txt) for 4/3 of the table memory.  The levels are built smallest first and each takes what the device
        // can spare at that moment: 128-B records while that leaves >= 60 % of the device free, packed records while that leaves >= 50 %
        // (the pass arenas of an n = 24 proof grow to ~110 GB, its self-checked / serial-steps forms to more: profiles/r06n_n24_classes.txt),
        // no table -- the table-less path -- below that.  A GPU shared by several parties
        // (threads mode) fills up as the parties build their sets, and the later / larger levels fall back by themselves.
        // ZKHOST_TABLE_REC=96 / 128 forces one layout for every level (A/B runs).
        static const int table_rec = [] {
            const char *e = std::getenv("ZKHOST_TABLE_REC");
            return e ? std::atoi(e) : 0;
        }();
        for (auto &lv : all) {  // largest levels last: the ones without a table simply use the table-less path
            size_t len = lv->len();
            if (len < 64 || len > (size_t(1) << table_max_log2)) continue;
            size_t fr = 0, tot = 0;
            be.check(zk_mem_info(be.handle(), &fr, &tot));
            const double copies = 16.0;  // (an upper bound of the table's window count at every width the library picks)
            auto left_after = [&](double rec_bytes) { return ((double)fr - copies * rec_bytes * (double)len) / (double)tot; };
            int rec = 96;
            if (table_rec == 96 || table_rec == 128) {
                rec = table_rec;
                if (left_after(rec) < 0.5) break;
            } else if (left_after(128.0) >= 0.6) {
                rec = 128;
            } else if (left_after(96.0) < 0.5) {
                break;
            }
            int rc = zk_srs_precompute_layout(be.handle(), lv->handle(), window_bits ? window_bits(len) : 0, rec);
            if (rc == ZK_ERR_OOM) break;
            be.check(rc);
        }
    }
    // dhyperplonk.rs:65-156 with a documented seed instead of StdRng::from_entropy(): tables from SplitMix64(seed ...)
    static PackedProvingParameters make(Ctx &be, size_t n, const PackedSharingParams &pp, uint64_t seed, uint64_t chal_seed = 0, bool window_tables = true,
                                        const std::function<int(size_t)> &window_bits = nullptr, size_t table_max_log2 = 25) {
        PackedProvingParameters pk;
        pk.n = n;
        uint64_t sd = 0x5CA1AB1Eull + 1000 * seed;
        for (auto &nl : layout(n, pp)) pk.put(be, nl.first, SplitMix64(++sd).fr_vec(nl.second));
        pk.set_challenges(SplitMix64(chal_seed ? chal_seed : ++sd).fr_vec(3 * n + 7));  // public values every party shares
        pk.finish_setup(be, pp, seed, window_tables, table_max_log2, window_bits);
        return pk;
    }
};

struct Transcript {
    std::vector<std::vector<Triple>> gate_proofs;          // the six gate sumchecks (:223-260)
    std::vector<std::pair<G1, Opening>> gate_commitments;  // (commitment, opening) of a, b, c, I_p, S1_p, S2_p
    std::vector<std::vector<Triple>> wiring_proofs;
    G1Vec wiring_commits;
    std::vector<Opening> wiring_opens;
};

namespace detail {
// step 2 of dhyperplonk (:262-514) == the body of dpermcheck (:992-1245), up to (not including) its one batched MSM pass:
// every sumcheck / fold / open-round kernel has run, every MSM of the step sits in `q`.  -> finalize(): the exchanges of the
// MSM results, filling the wiring lists in the reference's order.
inline std::function<void(Transcript &)> wiring_enqueue(size_t n, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, MsmQueue &q,
                                                        bool data_parallel, Timers *tm = nullptr) {
    auto mark = [tm](const char *w) {
        if (tm) tm->mark(w);
    };
    size_t l = pp.l, np = net.n_parties, M = size_t(1) << n, sbits = log2_floor(np);
    const PowersOfG &cc = pk.c_commitment, &dc = pk.d_commitment;
    const DevPtr &local_s_p = pk.T("local_s_p"), &local_s_l = pk.T("local_s_l");
    auto tr = std::make_shared<Transcript>();
    // 2.a (:268-294): every party broadcasts local_s; s = concatenation over parties (an all-gather that stays in HBM over RCCL)
    mark("wiring: begin");
    DevPtr s_dev = data_parallel ? pk.T("s_data_parallel") : net.all_gather_device(be, local_s_l, 32 * (4 * M / np / l));
    mark("2.a all-gather done");
    tr->wiring_proofs.push_back(c_sumcheck_product(be, s_dev, pk.T("V"), 4 * M / l, pk.challenge_r1, pp, net));  // 2.c
    mark("2.c c_sumcheck_product done");
    // 2.d: the two opens of V are independent -> their q_i commitments share one d_msm
    auto f_copen = c_open_many_q(be, q, cc, {pk.T("V"), pk.T("V")}, {4 * M / l, 4 * M / l}, {pk.challenge_r1, pk.challenge_r2}, pp, net);
    mark("2.d c_open_many_q (kernels) done");
    // 2.e (:322-340)
    size_t hlen = 4 * M / np;
    DevPtr num = be.fr_axpb(local_s_p, pk.T("sid_p"), pk.alpha, pk.beta, hlen);
    DevPtr den = be.fr_axpb(pk.T("eq_r1_p"), pk.T("ssigma_p"), pk.alpha, pk.beta, hlen);
    DevPtr h_p = be.fr_batch_div(num, den, hlen);
    mark("2.e num / den / h done");
    auto [sub, top] = d_acc_product(be, h_p, hlen, net);  // :342
    mark("2.e d_acc_product done");
    q.keep.push_back(sub.tree);  // v1x and the layer slices below are views into it
    DevPtr v1x = sub.tree.fr(hlen);
    auto [vx0, vx1] = be.fr_deinterleave(sub.tree, hlen);  // :344-359
    // :363-380 / :383-407: independent commits / opens
    std::vector<DevPtr> tabs8 = {pk.T("ssigma_p"), pk.T("sid_p"), h_p, num, den, v1x, vx0, vx1};
    std::vector<DevPtr> com_tabs = {local_s_p};
    com_tabs.insert(com_tabs.end(), tabs8.begin(), tabs8.end());
    std::vector<size_t> com_lens(9, hlen);
    auto f_dcommit = d_commit_many_q(be, q, dc, com_tabs, com_lens, net);  // 2.b, then :363-380
    std::vector<DevPtr> lay_tabs(com_tabs.begin(), com_tabs.begin() + 6);   // 2.d, then :383-407
    std::vector<size_t> lay_lens(6, hlen);
    std::vector<FrVec> lay_pts(6, pk.challenge_r2);
    // the 3 + 3 (n - s) d_sumcheck_products of 2.e are independent of each other: one batched local phase, one exchange
    std::vector<DsumcheckItem> dsp = {{den, pk.T("eq_r2_p"), hlen, pk.challenge_r2}, {h_p, den, hlen, pk.challenge_r2}, {num, pk.T("eq_r2_p"), hlen, pk.challenge_r2}};  // :411-413
    // 2.e.2 layered sumcheck + opens on halving slices (:417-478)
    DevPtr cur[4] = {v1x, vx0, vx1, pk.T("eq_r2_p")};
    size_t clen = hlen / 2;  // current_* = first half
    for (size_t i = 1; i + sbits <= n; ++i) {
        FrVec ch(pk.challenge_r2.begin() + i, pk.challenge_r2.end());
        dsp.push_back({cur[3], cur[0], clen, ch}), dsp.push_back({cur[3], cur[1], clen, ch}), dsp.push_back({cur[1], cur[2], clen, ch});
        for (int k = 0; k < 3; ++k) lay_tabs.push_back(cur[k]), lay_lens.push_back(clen), lay_pts.push_back(ch);
        for (auto &c : cur) c = c.fr(clen / 2);  // current = current[len/2..]
        clen /= 2;
    }
    mark("2.e lists built");
    auto f_dsp = d_sumcheck_product_many_q(be, dsp, net);
    mark("2.e d_sumcheck_product_many_q (kernels) done");
    auto f_dopen = d_open_many_q(be, q, dc, lay_tabs, lay_lens, lay_pts, net);
    mark("2.e d_open_many_q (kernels) done");
    // leader-only tail on the N_p-leaf top tree (:480-511)
    auto top_commits = std::make_shared<std::vector<std::function<G1()>>>();
    auto top_opens = std::make_shared<OpensInFlight>();
    auto top_proofs = std::make_shared<std::vector<std::vector<Triple>>>();
    bool has_top = top.has_value();
    if (has_top) {
        const FrVec &tt = *top;
        size_t half = tt.size() / 2;
        FrVec lv1x(tt.begin() + half, tt.end()), lvx0, lvx1;
        for (size_t i = 0; i < tt.size(); ++i) (i % 2 ? lvx1 : lvx0).push_back(tt[i]);
        FrVec chs(pk.challenge_r2.begin(), pk.challenge_r2.begin() + sbits);
        DevPtr d0 = be.to_device(lvx0), dd1 = be.to_device(lvx1), d1 = be.to_device(lv1x);
        for (auto &d : {d0, dd1, d1}) top_commits->push_back(commit_q(q, dc, d, half)), q.keep.push_back(d);
        *top_opens = open_many_q(be, q, dc, {d0, dd1, d1}, {half, half, half}, {chs, chs, chs});
        top_proofs->push_back(sumcheck_product(be, pk.T("eq_top"), d1, half, chs));
        top_proofs->push_back(sumcheck_product(be, pk.T("eq_top"), d0, half, chs));
        top_proofs->push_back(sumcheck_product(be, d0, dd1, half, chs));
    }
    mark("wiring: leader tail done");
    return [tr, f_dsp, f_copen, f_dcommit, f_dopen, top_commits, top_opens, top_proofs, has_top](Transcript &out) {
        out.wiring_proofs = tr->wiring_proofs;
        for (auto &p : f_dsp()) out.wiring_proofs.push_back(p);  // 2.e: after 2.c, before the leader-tree sumchecks (the reference's order)
        out.wiring_opens = f_copen();                           // 2.d
        out.wiring_commits = f_dcommit();                       // 2.b, then :363-380
        for (auto &o : f_dopen()) out.wiring_opens.push_back(o);
        if (has_top) {
            std::vector<Opening> to = top_opens->finish();
            for (size_t i = 0; i < 3; ++i) {  // (commit, open) per table, in the reference's order
                out.wiring_commits.push_back((*top_commits)[i]());
                out.wiring_opens.push_back(to[i]);
            }
            for (auto &p : *top_proofs) out.wiring_proofs.push_back(p);
        }
    };
}

// The same step with ALL its sumcheck-family kernels in the caller's batch (pipeline.hpp, ScQueue): phase A enqueues -- the
// exchanges that need no kernel result (2.a, the roots of the product tree) happen here --, the caller runs the batch, phase B
// (the returned function) performs the exchanges that need one and returns the finishing closure of wiring_enqueue.
inline AfterBatch<std::function<void(Transcript &)>> wiring_enqueue_sq(size_t n, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net,
                                                                       MsmQueue &q, ScQueue &sq, bool data_parallel, Timers *tm = nullptr) {
    auto mark = [tm](const char *w) {
        if (tm) tm->mark(w);
    };
    size_t l = pp.l, np = net.n_parties, M = size_t(1) << n, sbits = log2_floor(np);
    const PowersOfG &cc = pk.c_commitment, &dc = pk.d_commitment;
    const DevPtr &local_s_p = pk.T("local_s_p"), &local_s_l = pk.T("local_s_l");
    mark("wiring: begin");
    DevPtr s_dev = data_parallel ? pk.T("s_data_parallel") : net.all_gather_device(be, local_s_l, 32 * (4 * M / np / l));  // 2.a
    auto a_2c = c_sumcheck_product_many_sq(be, sq, {{s_dev, pk.T("V")}}, 4 * M / l, pk.challenge_r1, pp, net);              // 2.c
    auto a_copen = c_open_many_sq(be, sq, q, cc, {pk.T("V"), pk.T("V")}, {4 * M / l, 4 * M / l}, {pk.challenge_r1, pk.challenge_r2}, pp, net);  // 2.d
    size_t hlen = 4 * M / np;  // 2.e (:322-340)
    DevPtr num = be.fr_axpb(local_s_p, pk.T("sid_p"), pk.alpha, pk.beta, hlen);
    DevPtr den = be.fr_axpb(pk.T("eq_r1_p"), pk.T("ssigma_p"), pk.alpha, pk.beta, hlen);
    DevPtr h_p = be.fr_batch_div(num, den, hlen);
    auto [sub, top] = d_acc_product(be, h_p, hlen, net);  // :342
    mark("2.e product tree done");
    q.keep.push_back(sub.tree);
    DevPtr v1x = sub.tree.fr(hlen);
    auto [vx0, vx1] = be.fr_deinterleave(sub.tree, hlen);  // :344-359
    std::vector<DevPtr> com_tabs = {local_s_p, pk.T("ssigma_p"), pk.T("sid_p"), h_p, num, den, v1x, vx0, vx1};
    auto f_dcommit = d_commit_many_q(be, q, dc, com_tabs, std::vector<size_t>(9, hlen), net);  // 2.b, then :363-380
    std::vector<DevPtr> lay_tabs(com_tabs.begin(), com_tabs.begin() + 6);                      // 2.d, then :383-407
    std::vector<size_t> lay_lens(6, hlen);
    std::vector<FrVec> lay_pts(6, pk.challenge_r2);
    std::vector<DsumcheckItem> dsp = {{den, pk.T("eq_r2_p"), hlen, pk.challenge_r2}, {h_p, den, hlen, pk.challenge_r2}, {num, pk.T("eq_r2_p"), hlen, pk.challenge_r2}};  // :411-413
    DevPtr cur[4] = {v1x, vx0, vx1, pk.T("eq_r2_p")};
    size_t clen = hlen / 2;
    for (size_t i = 1; i + sbits <= n; ++i) {  // 2.e.2 layered sumcheck + opens on halving slices (:417-478)
        FrVec ch(pk.challenge_r2.begin() + i, pk.challenge_r2.end());
        dsp.push_back({cur[3], cur[0], clen, ch}), dsp.push_back({cur[3], cur[1], clen, ch}), dsp.push_back({cur[1], cur[2], clen, ch});
        for (int k = 0; k < 3; ++k) lay_tabs.push_back(cur[k]), lay_lens.push_back(clen), lay_pts.push_back(ch);
        for (auto &c : cur) c = c.fr(clen / 2);
        clen /= 2;
    }
    auto f_dsp = d_sumcheck_product_many_sq(be, sq, dsp, net);
    auto a_dopen = d_open_many_sq(be, sq, q, dc, lay_tabs, lay_lens, lay_pts, net);
    // leader-only tail on the N_p-leaf top tree (:480-511)
    auto top_commits = std::make_shared<std::vector<std::function<G1()>>>();
    auto top_opens = std::make_shared<OpensInFlight>();
    auto top_proofs = std::make_shared<std::vector<AfterBatch<std::vector<Triple>>>>();
    bool has_top = top.has_value();
    if (has_top) {
        const FrVec &tt = *top;
        size_t half = tt.size() / 2;
        FrVec lv1x(tt.begin() + half, tt.end()), lvx0, lvx1;
        for (size_t i = 0; i < tt.size(); ++i) (i % 2 ? lvx1 : lvx0).push_back(tt[i]);
        FrVec chs(pk.challenge_r2.begin(), pk.challenge_r2.begin() + sbits);
        DevPtr d0 = be.to_device(lvx0), dd1 = be.to_device(lvx1), d1 = be.to_device(lv1x);
        for (auto &d : {d0, dd1, d1}) top_commits->push_back(commit_q(q, dc, d, half)), q.keep.push_back(d);
        *top_opens = open_many_sq(sq, q, dc, {d0, dd1, d1}, {half, half, half}, {chs, chs, chs});
        top_proofs->push_back(sumcheck_product_sq(be, sq, pk.T("eq_top"), d1, half, chs));
        top_proofs->push_back(sumcheck_product_sq(be, sq, pk.T("eq_top"), d0, half, chs));
        top_proofs->push_back(sumcheck_product_sq(be, sq, d0, dd1, half, chs));
    }
    mark("wiring: everything enqueued");
    return [a_2c, a_copen, a_dopen, f_dsp, f_dcommit, top_commits, top_opens, top_proofs, has_top]() -> std::function<void(Transcript &)> {
        auto p2c = std::make_shared<std::vector<std::vector<Triple>>>(a_2c());  // pss2ss of 2.c
        auto f_copen = a_copen();                                              // pss2ss of the two opens of V
        auto f_dopen = a_dopen();                                              // the local open values, the leader's root opens
        return [p2c, f_dsp, f_copen, f_dcommit, f_dopen, top_commits, top_opens, top_proofs, has_top](Transcript &out) {
            out.wiring_proofs = *p2c;
            for (auto &p : f_dsp()) out.wiring_proofs.push_back(p);
            out.wiring_opens = f_copen();
            out.wiring_commits = f_dcommit();
            for (auto &o : f_dopen()) out.wiring_opens.push_back(o);
            if (has_top) {
                std::vector<Opening> to = top_opens->finish();
                for (size_t i = 0; i < 3; ++i) {
                    out.wiring_commits.push_back((*top_commits)[i]());
                    out.wiring_opens.push_back(to[i]);
                }
                for (auto &p : *top_proofs) out.wiring_proofs.push_back(p());
            }
        };
    };
}
}  // namespace detail

// dhyperplonk.rs:159-571 (data_parallel: dhyperplonk_data_parallel :573-960, which differs only at step 2.a -- `s` is local
// random data, no exchange, :603)
inline Transcript dhyperplonk(size_t n, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, Timers *tm_out = nullptr,
                              bool data_parallel = false, bool serial_steps = false) {
    size_t l = pp.l, M = size_t(1) << n, Ml = M / l;
    const PowersOfG &cc = pk.c_commitment, &dc = pk.d_commitment;
    Timers tm;
    if (tm_out) tm.keep_marks = tm_out->keep_marks;
    Transcript out;
    net.sync();
    tm.start("Distributed HyperPlonk");
    tm.mark("proof begin");

    // Step 1: commit (:198-215): both commit families in one batched MSM pass, started here and collected later
    tm.start("Commit");
    const char *names_c[3] = {"a_evals", "b_evals", "c_evals"}, *names_d[3] = {"I_p", "S1_p", "S2_p"};
    std::vector<DevPtr> tc, td;
    std::vector<size_t> lc, ld;
    for (auto x : names_c) tc.push_back(pk.T(x)), lc.push_back(pk.L(x));
    for (auto x : names_d) td.push_back(pk.T(x)), ld.push_back(pk.L(x));
    MsmQueue q(be);
    auto f_c = c_commit_q(be, q, cc, tc, lc, pp, net);
    auto f_d = d_commit_many_q(be, q, dc, td, ld, net);
    G1Vec com;
    auto collect_commit = [&] {
        com = f_c();
        G1Vec com_d = f_d();
        com.insert(com.end(), com_d.begin(), com_d.end());
    };
    std::vector<FrVec> pts3(3, pk.challenge);
    std::vector<Opening> ops;
    // Where the commit pass is started.  2 (default, one-batch schedule): right after the kernel batch of steps 2-4 -- the product tree
    // and the batch then run on an empty chip instead of queueing behind the pass, and the GPU works on the pass during the hand-offs
    // (n = 18: 27.5 -> 24.5 ms, 64 parties at n = 20: 22.8 -> 20.5 ms, neutral at n = 12, 16, 20 .. 24: profiles/r05z_late_commit2_ab.txt);
    // 0: at step 1, as the reference orders it; 1: together with the two long passes (ZKHOST_LATE_COMMIT)
    static const int late_commit = [] {
        const char *e = std::getenv("ZKHOST_LATE_COMMIT");
        const int v = e ? std::atoi(e) : 2;
        return (v == 0 || v == 1) ? v : 2;  // anything else is the default (an unknown value must not leave the commit pass unstarted)
    }();
    static const bool one_batch = [] {  // the sumcheck-family kernels of steps 2-4 as ONE batch (pipeline.hpp ScQueue); ZKHOST_ONE_BATCH=0: a batch per call
        const char *e = std::getenv("ZKHOST_ONE_BATCH");
        return !e || std::atoi(e) != 0;
    }();
    const bool commit_at_step_1 = late_commit == 0 || (late_commit == 2 && !one_batch);
    if (serial_steps) {
        q.run();
        collect_commit();
    } else if (commit_at_step_1) {
        q.start();
    }
    tm.mark("commit pass started");
    tm.end();

    if (one_batch && !serial_steps) {
        ScQueue sq(be);
        MsmQueue q_w(be), q_o(be);
        tm.start("Gate identity");
        DevPtr sum_ab = be.fr_add(pk.T("a_evals"), pk.T("b_evals"), Ml);  // :233-238
        DevPtr sum_ci = be.fr_sub(pk.T("I"), pk.T("c_evals"), Ml);        // -c + I  :251-256
        auto a_gate = c_sumcheck_product_many_sq(be, sq, {{pk.T("eq"), pk.T("S1")}, {pk.T("S1"), sum_ab}, {pk.T("eq"), pk.T("S2")}, {pk.T("a_evals"), pk.T("b_evals")},
                                                          {pk.T("S2"), pk.T("a_evals")}, {pk.T("eq"), sum_ci}}, Ml, pk.challenge, pp, net);
        tm.end();
        tm.start("Wire identity");
        auto b_wiring = detail::wiring_enqueue_sq(n, pk, pp, be, net, q_w, sq, data_parallel, &tm);
        auto a_co = c_open_many_sq(be, sq, q_o, cc, tc, lc, pts3, pp, net);  // the kernel phase of step 4 (:517-553) rides in the same batch
        auto a_do = d_open_many_sq(be, sq, q_o, dc, td, ld, pts3, net);
        sq.run();
        tm.mark("the batch ran");
        if (late_commit == 2) q.start();
        out.gate_proofs = a_gate();
        auto finalize_wiring = b_wiring();
        auto f_co = a_co();
        auto f_do = a_do();
        tm.mark("hand-offs done");
        if (late_commit == 1) q.start();
        q_w.start();
        tm.mark("wiring pass started");
        q_o.start();
        tm.mark("open pass started");
        q.finish();
        collect_commit();
        tm.mark("commit pass collected");
        tm.end();
        tm.start("Open");
        q_w.finish();
        tm.mark("wiring pass finished");
        finalize_wiring(out);
        tm.mark("wiring finalized");
        q_o.finish();
        tm.mark("open pass finished");
        ops = f_co();
        std::vector<Opening> ops_d = f_do();
        ops.insert(ops.end(), ops_d.begin(), ops_d.end());
        tm.end();
        for (size_t i = 0; i < 6; ++i) out.gate_commitments.push_back({com[i], ops[i]});
        tm.end();
        if (tm_out) *tm_out = tm;
        return out;
    }

    // Step 3: gate identity (:223-260): the six sumchecks are independent -- one batched phase 1, then the hand-offs in order
    tm.start("Gate identity");
    DevPtr sum_ab = be.fr_add(pk.T("a_evals"), pk.T("b_evals"), Ml);  // :233-238
    DevPtr sum_ci = be.fr_sub(pk.T("I"), pk.T("c_evals"), Ml);        // -c + I  :251-256
    out.gate_proofs = c_sumcheck_product_many(be, {{pk.T("eq"), pk.T("S1")}, {pk.T("S1"), sum_ab}, {pk.T("eq"), pk.T("S2")}, {pk.T("a_evals"), pk.T("b_evals")},
                                                   {pk.T("S2"), pk.T("a_evals")}, {pk.T("eq"), sum_ci}}, Ml, pk.challenge, pp, net);
    tm.mark("gate sumchecks done");
    tm.end();

    MsmQueue q_w(be), q_o(be);
    if (serial_steps) {
        // every MSM pass runs to completion inside the step that owns it: the timers cover what the reference's labels cover
        // (same transcript; the measurement form, `hyperplonk --serial-rep`)
        tm.start("Wire identity");
        auto finalize_wiring = detail::wiring_enqueue(n, pk, pp, be, net, q_w, data_parallel);
        q_w.run();
        finalize_wiring(out);
        tm.end();
        tm.start("Open");
        auto f_co = c_open_many_q(be, q_o, cc, tc, lc, pts3, pp, net);
        auto f_do = d_open_many_q(be, q_o, dc, td, ld, pts3, net);
        q_o.run();
        ops = f_co();
        std::vector<Opening> ops_d = f_do();
        ops.insert(ops.end(), ops_d.begin(), ops_d.end());
        tm.end();
    } else {
        // Step 2: wiring identity (shared with dpermcheck).  The kernel phase of the Open step (:517-553) depends on nothing the
        // wiring step produces, so it runs BEFORE the wiring pass is started: both passes are then in flight back to back.
        // (The labels below are therefore OVERLAPPED sections, not the reference's steps: only the total is comparable.)
        tm.start("Wire identity");
        auto finalize_wiring = detail::wiring_enqueue(n, pk, pp, be, net, q_w, data_parallel, &tm);
        auto f_co = c_open_many_q(be, q_o, cc, tc, lc, pts3, pp, net);
        auto f_do = d_open_many_q(be, q_o, dc, td, ld, pts3, net);
        tm.mark("open-step kernels done");
        if (!commit_at_step_1) q.start();
        q_w.start();
        tm.mark("wiring pass started");
        q_o.start();
        tm.mark("open pass started");
        q.finish();  // (host: exchange + point combinations of step 1, beside the passes on the GPU)
        collect_commit();
        tm.mark("commit pass collected");
        tm.end();

        // Open (:517-553): collection of both passes
        tm.start("Open");
        q_w.finish();
        tm.mark("wiring pass finished");
        finalize_wiring(out);
        tm.mark("wiring finalized");
        q_o.finish();
        tm.mark("open pass finished");
        ops = f_co();
        std::vector<Opening> ops_d = f_do();
        ops.insert(ops.end(), ops_d.begin(), ops_d.end());
        tm.end();
    }
    for (size_t i = 0; i < 6; ++i) out.gate_commitments.push_back({com[i], ops[i]});
    tm.end();
    if (tm_out) *tm_out = tm;
    return out;
}

// hyperplonk/src/dhyperplonk.rs:962-1247: the distributed permutation check alone (= step 2 of dhyperplonk)
inline Transcript dpermcheck(size_t n, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, Timers *tm_out = nullptr) {
    Timers tm;
    Transcript out;
    net.sync();
    tm.start("Distributed Permcheck");
    MsmQueue q(be);
    ScQueue sq(be);  // the step's sumcheck-family kernels as one batch (pipeline.hpp)
    auto after = detail::wiring_enqueue_sq(n, pk, pp, be, net, q, sq, false);
    sq.run();
    auto fin = after();
    q.run();
    fin(out);
    tm.end();
    if (tm_out) *tm_out = tm;
    return out;
}

// hyperplonk/src/dhyperplonk.rs:1249-1385: the collaborative (packed) permutation check: num / den maps, c_commit / c_open of
// the public wires, and per polynomial the masked product tree (c_acc_product_and_share) with its commits, opens and three
// product sumchecks.  masks: "mask", "unmask0..2" tables of 4 (2^n / l) Fr (PackedProvingParameters::new :124-127).
inline Transcript cpermcheck(size_t n, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, Timers *tm_out = nullptr) {
    size_t G4 = 4 * ((size_t(1) << n) / pp.l);  // gate_count * 4 with gate_count = 2^n / l (:1270)
    const PowersOfG &cc = pk.c_commitment;
    Timers tm;
    Transcript out;
    net.sync();
    tm.start("Collaborative Permcheck");
    DevPtr num = be.fr_axpb(pk.T("V"), pk.T("sid"), pk.alpha, pk.beta, G4);          // :1277-1279
    DevPtr den = be.fr_axpb(pk.T("eq_r1"), pk.T("ssigma"), pk.alpha, pk.beta, G4);  // :1280-1282
    // The reference runs 10 c_commit, 12 c_open and 6 c_sumcheck_product one after the other (:1289-1375); none of them feeds another on
    // the device -- only the two masked product trees produce tables the rest reads.  So: the trees first, then ALL commitments and
    // quotient commitments as ONE MSM pass (22 x 4 G scalar-muls: the pass runs at the rate of a large MSM instead of 22 blocking
    // ones) and the fold rounds of the 12 opens + the six product sumchecks as ONE kernel batch.  The second open of num / den (:1371-1375)
    // is the same table at the same point as the first: c_open_many_sq adds its kernels and MSM items once.  Outputs keep the
    // reference's positions.
    static const bool serial = std::getenv("ZKHOST_CPERM_SERIAL") && std::atoi(std::getenv("ZKHOST_CPERM_SERIAL"));  // A/B: the call-by-call form
    if (serial) {
        auto commit_open = [&](const DevPtr &tab) {
            out.wiring_commits.push_back(c_commit(be, cc, {tab}, {G4}, pp, net)[0]);
            out.wiring_opens.push_back(c_open(be, cc, tab, G4, pk.challenge_r1, pp, net));
        };
        commit_open(pk.T("ssigma")), commit_open(pk.T("sid"));  // :1289-1308
        for (const DevPtr &ev : {num, den}) {
            auto sh = c_acc_product_and_share(be, ev, pk.T("mask"), pk.T("unmask0"), pk.T("unmask1"), pk.T("unmask2"), G4, pp, net);
            if (sh[0].len != G4 || sh[1].len != G4 || sh[2].len != G4) throw ZkError(ZK_ERR_INVALID, "cpermcheck: share vectors of unexpected length");
            commit_open(ev), commit_open(sh[0].buf), commit_open(sh[1].buf), commit_open(sh[2].buf);  // :1324-1363
            out.wiring_proofs.push_back(c_sumcheck_product(be, pk.T("eq_r1"), sh[2].buf, G4, pk.challenge_r1, pp, net));  // :1365-1369
            out.wiring_proofs.push_back(c_sumcheck_product(be, pk.T("eq_r1"), sh[0].buf, G4, pk.challenge_r1, pp, net));
            out.wiring_proofs.push_back(c_sumcheck_product(be, sh[0].buf, sh[1].buf, G4, pk.challenge_r1, pp, net));
            out.wiring_opens.push_back(c_open(be, cc, ev, G4, pk.challenge_r1, pp, net));  // :1371-1375
        }
    } else {
        std::vector<DevPtr> com_tabs{pk.T("ssigma"), pk.T("sid")}, open_tabs{pk.T("ssigma"), pk.T("sid")};  // :1289-1308
        std::vector<std::pair<DevPtr, DevPtr>> pairs;
        for (const DevPtr &ev : {num, den}) {
            auto sh = c_acc_product_and_share(be, ev, pk.T("mask"), pk.T("unmask0"), pk.T("unmask1"), pk.T("unmask2"), G4, pp, net);
            if (sh[0].len != G4 || sh[1].len != G4 || sh[2].len != G4) throw ZkError(ZK_ERR_INVALID, "cpermcheck: share vectors of unexpected length");
            for (const DevPtr &t : {ev, sh[0].buf, sh[1].buf, sh[2].buf}) com_tabs.push_back(t), open_tabs.push_back(t);  // :1324-1363
            pairs.push_back({pk.T("eq_r1"), sh[2].buf}), pairs.push_back({pk.T("eq_r1"), sh[0].buf}), pairs.push_back({sh[0].buf, sh[1].buf});  // :1365-1369
            open_tabs.push_back(ev);  // :1371-1375
        }
        MsmQueue q(be);
        ScQueue sq(be);
        auto f_com = c_commit_q(be, q, cc, com_tabs, std::vector<size_t>(com_tabs.size(), G4), pp, net);
        auto a_open = c_open_many_sq(be, sq, q, cc, open_tabs, std::vector<size_t>(open_tabs.size(), G4), std::vector<FrVec>(open_tabs.size(), pk.challenge_r1), pp, net);
        auto a_sc = c_sumcheck_product_many_sq(be, sq, pairs, G4, pk.challenge_r1, pp, net);
        sq.run();
        auto f_open = a_open();
        out.wiring_proofs = a_sc();
        q.run();
        out.wiring_commits = f_com();
        out.wiring_opens = f_open();
    }
    tm.end();
    if (tm_out) *tm_out = tm;
    return out;
}

}  // namespace zkhost
