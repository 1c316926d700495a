// The C++ host's counterpart of the reference's hyperplonk/examples/{hyperplonk,bench_hyperplonk,bench_hyperplonk_dataparallel,
// bench_dpermcheck,bench_cpermcheck}.rs: build the synthetic parameter set (PackedProvingParameters::new, dhyperplonk.rs:65-156),
// run the collaborative proof on the GPU(s), print the reference's timer labels and its `Comm: (up, down)` line (:564).
//
//   hyperplonk --l L --n N [--mode leader|threads|rccl] [--which dhyperplonk|data-parallel|dpermcheck|cpermcheck] [--reps R] [--no-tables] [--table-max LOG2]
//              [--digest] [--dump PREFIX] [--check] [--tamper] [--serial-rep] [--arena-plan FILE]
//     leader   party 0 on the no-`comm` echo net (the reference's `-F leader` build: one party's full work; default)
//     threads  all 8 l parties as threads of this process, one ctx each, exchanges through host memory (LocalTestNet); the
//              parties share the visible GPUs round-robin
//     rccl     all 8 l parties as threads, party p on GPU p, exchanges over RCCL / xGMI inside the ctx (zk_comm_init_all);
//              needs 8 l GPUs (--share-gpus: fewer, for runs against the test double tests/native/fake_rccl only)
//     --check  SELF-CHECKING run, no Python and no oracle in the loop (zkhost/verify.hpp): one more run with the operands of every
//              product sumcheck traced; every transcript chain (dsumcheck.rs:558-588) is pinned at both ends by values from
//              kernels the product sumcheck does not use (every party's values for the leader's d_ chains), the c_ tails are
//              recomputed from pss2ss of independently folded values, sampled commits / opens must equal the one-call-at-a-time
//              forms, the repetitions and the traced run must agree bit for bit, and a copy of the transcript with one limb
//              flipped must be rejected under exactly its label.  Prints one `check: ...` line per party; exit code 3 when any
//              party fails.  --tamper flips that limb in the transcript under test instead (the run must then FAIL: exit 3).
//     --dump PREFIX  EVERY party writes the transcript of its last repetition to PREFIX.party<p>.bin (raw limbs behind u64 counts, in
//              the reference's order): tests/test_protocol_oracle.py compares it position by position with the oracle's straight-line
//              statement of the reference's call sequence
//     --arena-plan FILE  (leader mode) the ctx's arena plan (zk_arena_plan_export / _import): imported after the setup when FILE exists -- the
//              first proof then allocates nothing --, written after the repetitions.  Keep it beside the proving key: it depends on
//              (n, l, which) only
//     --serial-rep  after the timed repetitions, one more proof with every MSM pass run to completion inside the step that owns
//              it (`End(serial):` lines): the per-step timers of the timed repetitions are OVERLAPPED sections
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "sha256.hpp"
#include "zkhost/verify.hpp"

using namespace zkhost;

struct Args {
    size_t l = 1, n = 12, reps = 3, table_max = 25;
    std::string mode = "leader", which = "dhyperplonk", dump, arena_plan;
    bool tables = true, digest = false, check = false, tamper = false, serial_rep = false, marks = false, share_gpus = false;
};

static std::atomic<int> g_failed{0};  // parties whose self-check failed

static Transcript run_once(const Args &a, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, Timers &tm, bool serial_steps = false) {
    if (a.which == "cpermcheck") return cpermcheck(a.n, pk, pp, be, net, &tm);
    if (a.which == "dpermcheck") return dpermcheck(a.n, pk, pp, be, net, &tm);
    return dhyperplonk(a.n, pk, pp, be, net, &tm, a.which == "data-parallel", serial_steps);
}

// SHA-256 over the transcript in the reference's order: gate proofs, (commitment, value, proofs) of the six gate openings, wiring
// proofs, wiring commitments, (value, proofs) of the wiring openings -- raw limbs, as tests/test_host_cpp.py hashes the Python host's
static std::string transcript_digest(const Transcript &t) {
    Sha256 h;
    for (auto &p : t.gate_proofs) h.update(p.data(), 96 * p.size());
    for (auto &c : t.gate_commitments) h.update(c.first.data(), 144), h.update(c.second.value.v, 32), h.update(c.second.proofs.data(), 144 * c.second.proofs.size());
    for (auto &p : t.wiring_proofs) h.update(p.data(), 96 * p.size());
    h.update(t.wiring_commits.data(), 144 * t.wiring_commits.size());
    for (auto &o : t.wiring_opens) h.update(o.value.v, 32), h.update(o.proofs.data(), 144 * o.proofs.size());
    return h.hex();
}

// --dump: u64 counts + raw limbs, lists in the reference's order (the reader is tests/test_protocol_oracle.py::_read_dump)
static void dump_transcript(const Transcript &t, const std::string &path) {
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("--dump: cannot open " + path);
    auto u64 = [f](uint64_t v) { std::fwrite(&v, 8, 1, f); };
    auto proofs = [&](const std::vector<std::vector<Triple>> &ps) {
        u64(ps.size());
        for (auto &p : ps) u64(p.size()), std::fwrite(p.data(), 96, p.size(), f);
    };
    auto opening = [&](const Opening &o) {
        std::fwrite(o.value.v, 32, 1, f);
        u64(o.proofs.size());
        std::fwrite(o.proofs.data(), 144, o.proofs.size(), f);
    };
    proofs(t.gate_proofs);
    u64(t.gate_commitments.size());
    for (auto &c : t.gate_commitments) std::fwrite(c.first.data(), 144, 1, f), opening(c.second);
    proofs(t.wiring_proofs);
    u64(t.wiring_commits.size());
    std::fwrite(t.wiring_commits.data(), 144, t.wiring_commits.size(), f);
    u64(t.wiring_opens.size());
    for (auto &o : t.wiring_opens) opening(o);
    std::fclose(f);
}

// flip one limb of one t2 in the middle of a transcript every party holds (gate[3], or the first wiring transcript of the
// drivers without a gate step) -> its label
static std::string flip_one_limb(Transcript &t) {
    bool gate = t.gate_proofs.size() > 3;
    std::vector<Triple> &pr = gate ? t.gate_proofs[3] : t.wiring_proofs.at(0);
    pr.at(pr.size() / 2)[2].v[0] ^= 1;
    return gate ? "gate[3]" : "wiring[0]";
}

// --check (see the header comment); collective: every party runs it in lock step
static void self_check(const Args &a, const PackedProvingParameters &pk, const PackedSharingParams &pp, Ctx &be, Net &net, const std::vector<std::string> &digests) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<ScTrace> trace;
    Timers tm;
    be.sc_trace = &trace;
    Transcript t = run_once(a, pk, pp, be, net, tm);
    be.sc_trace = nullptr;
    CheckReport rep;
    for (auto &d : digests)
        if (d != digests[0]) rep.bad.push_back("repetitions disagree");
    if (!digests.empty() && transcript_digest(t) != digests.back()) rep.bad.push_back("the traced (anchored) run differs from the timed ones");
    ProductAnchors anchors = gather_product_anchors(be, trace, pp, net);
    trace.clear();
    if (a.tamper) flip_one_limb(t);
    check_product_transcripts(anchors, t, rep);
    bool full = a.which == "dhyperplonk" || a.which == "data-parallel";
    if (full) {
        check_dhyperplonk_shape(a.n, t, net, rep);
        check_dhyperplonk_recompute(a.n, pk, pp, be, net, t, rep);
    }
    if (a.which == "cpermcheck") check_cpermcheck_recompute(a.n, pk, pp, be, net, t, rep);
    // the check has teeth: one limb off in one t2 is rejected, under its own label and no other
    std::string teeth = "skipped (--tamper)";
    if (!a.tamper) {
        Transcript broken = t;
        std::string label = flip_one_limb(broken);
        CheckReport r2;
        check_product_transcripts(anchors, broken, r2);
        bool rejected = r2.bad.size() == 1 && r2.bad[0] == label;
        teeth = rejected ? "rejected as " + label : "NOT rejected";
        if (!rejected) rep.bad.push_back("a transcript with one limb flipped was not rejected as " + label);
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::string msg = rep.ok() ? "ok" : "FAILED";
    for (auto &b : rep.bad) msg += " [" + b + "]";
    static std::mutex out;
    std::lock_guard<std::mutex> lk(out);
    std::printf("check: party %zu of %zu %s -- anchored, %zu transcripts pinned at both ends, %zu c_ tails, %zu recomputed commits / opens, flipped limb %s (%.3f s)\n",
                (size_t)net.party_id, (size_t)net.n_parties, msg.c_str(), rep.transcripts, rep.closing_rows, rep.recomputed, teeth.c_str(), secs);
    if (!rep.ok()) ++g_failed;
}

static void party(const Args &a, const PackedSharingParams &pp, Ctx &be, Net &net) {
    size_t p = net.party_id;
    auto t0 = std::chrono::steady_clock::now();
    // per-party tables, shared public challenges (the reference's local mode clones ONE parameter set, mpc-net/src/multi.rs:344)
    // experiment hook: ZK_TABLE_POLICY="lo:hi:delta,..." shifts the table width of the levels with lo <= log2(len) <= hi
    std::function<int(size_t)> policy = nullptr;
    if (const char *e = std::getenv("ZK_TABLE_POLICY")) {
        std::string spec = e;
        policy = [spec](size_t len) {
            int lg = (int)log2_floor(len), c = lg <= 10 ? 12 : (lg <= 14 ? 14 : (lg == 15 ? 17 : (lg <= 18 ? 17 : lg - 1)));  // the library's pick (csrc/zk_msm.hip msm_pick_window_full)
            const char *q = spec.c_str();
            int lo, hi, d, used;
            while (std::sscanf(q, "%d:%d:%d%n", &lo, &hi, &d, &used) == 3) {
                if (lg >= lo && lg <= hi) c += d;
                q += used;
                if (*q == ',') ++q;
            }
            return std::max(4, std::min(20, c));
        };
    }
    PackedProvingParameters pk = PackedProvingParameters::make(be, a.n, pp, 100 + p, 4242, a.tables, policy, a.table_max);
    if (a.which == "cpermcheck") {
        size_t G4 = 4 * ((size_t(1) << a.n) / pp.l);
        const char *names[4] = {"mask", "unmask0", "unmask1", "unmask2"};
        for (size_t i = 0; i < 4; ++i) pk.put(be, names[i], SplitMix64(977 * (100 + p) + 50 + i).fr_vec(G4));
    }
    be.sync();
    double setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (net.is_leader()) {
        size_t fr = 0, tot = 0;
        if (!zk_mem_info(be.handle(), &fr, &tot)) std::printf("setup %.3f s; HBM after setup: %.1f GiB free of %.1f GiB\n", setup, fr / 1073741824.0, tot / 1073741824.0);
    }
    if (!a.arena_plan.empty() && net.n_parties > 0 && a.mode == "leader") {
        uint64_t plan[ZK_ARENA_PLAN_WORDS];
        if (std::FILE *f = std::fopen(a.arena_plan.c_str(), "rb")) {
            bool ok = std::fread(plan, 8, ZK_ARENA_PLAN_WORDS, f) == ZK_ARENA_PLAN_WORDS;
            std::fclose(f);
            auto t1 = std::chrono::steady_clock::now();
            if (ok) be.check(zk_arena_plan_import(be.handle(), plan));
            be.sync();
            std::printf("arena plan %s: %s (%.3f s)\n", a.arena_plan.c_str(), ok ? "imported" : "unreadable, ignored", std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
        }
    }
    std::vector<std::string> digests;
    std::vector<double> totals;
    for (size_t r = 0; r < a.reps; ++r) {
        Timers tm;
        tm.keep_marks = a.marks;
        uint64_t up0 = net.upload, down0 = net.download;
        Transcript t = run_once(a, pk, pp, be, net, tm);
        if (a.check) digests.push_back(transcript_digest(t));
        if (!a.dump.empty() && r + 1 == a.reps) dump_transcript(t, a.dump + ".party" + std::to_string(p) + ".bin");
        for (auto &kv : tm.t)
            if (kv.first.rfind("Distributed", 0) == 0 || kv.first.rfind("Collaborative", 0) == 0) totals.push_back(kv.second);
        if (net.is_leader()) {
            std::printf("rep %zu (setup %.3f s): proofs %zu + %zu, commitments %zu + %zu, openings %zu\n", r, setup, t.gate_proofs.size(), t.wiring_proofs.size(),
                        t.gate_commitments.size(), t.wiring_commits.size(), t.wiring_opens.size());
            for (auto &kv : tm.t) std::printf("  End: %-28s %.6f s\n", kv.first.c_str(), kv.second);
            std::printf("Comm: (%llu, %llu)\n", (unsigned long long)(net.upload - up0), (unsigned long long)(net.download - down0));
            if (a.digest) std::printf("transcript sha256 %s\n", transcript_digest(t).c_str());
            for (auto &m : tm.marks) std::printf("  mark %9.3f ms  %s\n", m.second * 1e3, m.first.c_str());
        }
    }
    if (net.is_leader()) {
        // (the MSM / sumcheck arenas stay with the ctx between proofs: free memory now = the device minus this run's whole footprint;
        // with one party per GPU -- `--mode rccl` on an 8-GPU node -- that footprint is the per-GPU figure of the configuration)
        size_t fr = 0, tot = 0;
        if (a.reps && !zk_mem_info(be.handle(), &fr, &tot)) std::printf("HBM after the proofs: %.1f GiB free of %.1f GiB (tables, parameter set and arenas resident: %.1f GiB in use on this device)\n", fr / 1073741824.0, tot / 1073741824.0, (tot - fr) / 1073741824.0);
    }
    if (!a.arena_plan.empty() && a.mode == "leader" && a.reps) {
        uint64_t plan[ZK_ARENA_PLAN_WORDS];
        be.check(zk_arena_plan_export(be.handle(), plan));
        if (std::FILE *f = std::fopen(a.arena_plan.c_str(), "wb")) {
            std::fwrite(plan, 8, ZK_ARENA_PLAN_WORDS, f);
            std::fclose(f);
        }
    }
    if (net.is_leader() && totals.size() >= 3) {
        // (the first proof of a process also sizes the library's arenas: left out)
        std::vector<double> v(totals.begin() + 1, totals.end());
        std::sort(v.begin(), v.end());
        std::printf("proofs after the first: min %.6f  median %.6f  max %.6f s over %zu\n", v.front(), v[v.size() / 2], v.back(), v.size());
    }
    if (net.is_leader() && a.reps && (a.which == "dhyperplonk" || a.which == "data-parallel"))
        std::printf("note: 'Commit' / 'Wire identity' / 'Open' above are OVERLAPPED sections (a step's MSM pass is started asynchronously and collected later; the Open step's "
                    "kernels run inside 'Wire identity'): they are not the reference's phases of the same name -- only 'Distributed HyperPlonk' is; --serial-rep prints the per-step form\n");
    if (a.serial_rep && (a.which == "dhyperplonk" || a.which == "data-parallel")) {
        // one more proof with every MSM pass run to completion inside its own step: per-step timers that cover what the
        // reference's labels cover (the timed repetitions above overlap their steps); same transcript
        Timers tm;
        Transcript t = run_once(a, pk, pp, be, net, tm, true);  // (the first blocking pass of the process sizes the ctx's own arenas:
        t = run_once(a, pk, pp, be, net, tm = Timers(), true);  //  the second run is the steady state, like the timed repetitions)
        if (a.check) digests.push_back(transcript_digest(t));
        if (net.is_leader()) {
            std::printf("serial-steps rep:\n");
            for (auto &kv : tm.t) std::printf("  End(serial): %-28s %.6f s\n", kv.first.c_str(), kv.second);
            if (a.digest) std::printf("transcript sha256 %s\n", transcript_digest(t).c_str());
        }
    }
    if (a.check) self_check(a, pk, pp, be, net, digests);
}

int main(int argc, char **argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value for %s\n", k.c_str());
                std::exit(64);
            }
            return argv[++i];
        };
        if (k == "--l") a.l = std::strtoull(val(), nullptr, 10);
        else if (k == "--n") a.n = std::strtoull(val(), nullptr, 10);
        else if (k == "--reps") a.reps = std::strtoull(val(), nullptr, 10);
        else if (k == "--mode") a.mode = val();
        else if (k == "--which") a.which = val();
        else if (k == "--no-tables") a.tables = false;
        else if (k == "--digest") a.digest = true;
        else if (k == "--dump") a.dump = val();
        else if (k == "--arena-plan") a.arena_plan = val();
        else if (k == "--check") a.check = true;
        else if (k == "--serial-rep") a.serial_rep = true;
        else if (k == "--marks") a.marks = true;  // diagnostics: host time stamps of the calls inside a proof
        else if (k == "--share-gpus") a.share_gpus = true;  // --mode rccl with fewer GPUs than parties: only the test double of librccl accepts it
        else if (k == "--tamper") a.check = a.tamper = true;
        else if (k == "--table-max") a.table_max = std::strtoull(val(), nullptr, 10);
        else if (k == "--table-rec") setenv("ZKHOST_TABLE_REC", val(), 1);  // 128: one G1 table record per cache line (4/3 of the table memory)
        else {
            std::fprintf(stderr, "usage: hyperplonk --l L --n N [--mode leader|threads|rccl] [--which dhyperplonk|data-parallel|dpermcheck|cpermcheck] [--reps R] [--no-tables] [--table-max LOG2] [--table-rec 96|128] [--digest] [--dump PREFIX] [--check] [--tamper] [--serial-rep] [--arena-plan FILE]\n");
            return 64;
        }
    }
    int ngpu = zk_device_count();
    if (ngpu <= 0) {
        std::fprintf(stderr, "hyperplonk: no GPU visible -- this host has no CPU fallback (zk_device_count = %d)\n", ngpu);
        return 2;
    }
    try {
        PackedSharingParams pp(a.l);
        if (a.n < log2_floor(pp.n) + 1) throw std::invalid_argument("n too small for this party count");
        if (a.mode == "leader") {
            Ctx be(0);
            LeaderEchoNet net(pp.n);
            party(a, pp, be, net);
        } else if (a.mode == "threads") {
            LocalTestNet::simulate_network_round(pp.n, [&](size_t p, LocalTestNet &net) {
                Ctx be((int)(p % (size_t)ngpu));
                party(a, pp, be, net);
            });
        } else if (a.mode == "rccl") {
            // (--share-gpus: the real RCCL refuses two ranks on one device; tests/native/fake_rccl -- a test double first in
            // LD_LIBRARY_PATH -- does not, and lets the whole RCCL path of the library and of this host run on a one-GPU box)
            if ((size_t)ngpu < pp.n && !a.share_gpus) throw std::invalid_argument("--mode rccl needs one GPU per party (" + std::to_string(pp.n) + "), found " + std::to_string(ngpu));
            std::vector<std::unique_ptr<Ctx>> ctxs;
            std::vector<zk_ctx *> raw;
            for (size_t p = 0; p < pp.n; ++p) ctxs.push_back(std::make_unique<Ctx>((int)(p % (size_t)ngpu))), raw.push_back(ctxs.back()->handle());
            ctxs[0]->check(zk_comm_init_all(raw.data(), (int)pp.n));
            // every party on its own host thread (the collectives block until all parties have entered them)
            LocalTestNet::simulate_network_round(pp.n, [&](size_t p, LocalTestNet &) {
                try {
                    RcclNet net(*ctxs[p]);
                    party(a, pp, *ctxs[p], net);
                } catch (...) {
                    // a party that fails outside zk_d_msm (which carries its own status words) -- an out-of-memory arena, say -- must not
                    // leave its peers waiting inside a collective (include/zkhip.h, zk_comm_abort)
                    zk_comm_abort(ctxs[p]->handle());
                    throw;
                }
            });
        } else {
            throw std::invalid_argument("unknown --mode " + a.mode);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "hyperplonk: %s\n", e.what());
        return 1;
    }
    if (g_failed.load()) {
        std::fprintf(stderr, "hyperplonk: the self-check failed on %d part%s\n", g_failed.load(), g_failed.load() == 1 ? "y" : "ies");
        return 3;
    }
    return 0;
}
