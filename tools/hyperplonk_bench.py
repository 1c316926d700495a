#!/usr/bin/env python3
"""
End-to-end collaborative HyperPlonk (hyperplonk/src/dhyperplonk.rs) on MI355X.

  python tools/hyperplonk_bench.py --n 16                       # `leader` mode: party 0 alone, no-comm echo net (config 1 style)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/hyperplonk_bench.py --nvars 20   # l = 1, 8 parties = 8 GPUs

Prints the reference's timer sections (Commit / Gate identity / Wire identity / Open / total)
and the Comm: (up, down) byte counters for the leader.  Every run is SELF-CHECKING (unless --no-check), in every mode
(leader-echo, ranks, party threads): all sumcheck transcripts must pass their verifier chains (zkhip.verify,
dsumcheck.rs:541-588) ANCHORED at both ends -- the claim sum f g and the final evaluation f(r) g(r) of every chain come from
kernels the product sumcheck does not use (an extra traced run, outside the timed repetitions; the leader's d_ chains use
every party's values); the closing rows of the c_sumcheck_products are compared with pss2ss of independently folded values;
sampled commits / opens must equal a one-call-at-a-time recomputation, and the timed repetitions must reproduce the same
transcript bit for bit; a failed check exits non-zero.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))


def digest(res):
    import hashlib

    import numpy as np

    h = hashlib.sha256()

    def feed(x):
        if isinstance(x, np.ndarray):
            h.update(np.ascontiguousarray(x, dtype=np.uint64).tobytes())
        elif isinstance(x, (list, tuple)):
            for e in x:
                feed(e)

    feed(res)
    return h.hexdigest()


def traced_run(n, pk, pp, ctx, net, run_seed, data_parallel):
    """one more run with the operands of every product sumcheck recorded -> (result, this party's anchor values, closing rows ok)"""
    from zkhip.hyperplonk import dhyperplonk
    from zkhip.verify import check_closing_rows, trace_anchor_values

    ctx.sc_trace = []
    res, _ = dhyperplonk(n, pk, pp, ctx, net, seed=run_seed, data_parallel=data_parallel)
    trace, ctx.sc_trace = ctx.sc_trace, None
    values = trace_anchor_values(ctx, trace)
    del trace
    closing = check_closing_rows(values[:7], list(res[0][0]) + [res[1][0][0]], pp, net)
    return res, values, closing


def anchored_failures(n, res_sc, chal, values_by_party, me, n_parties, leader, echo):
    """verifier chains of one party's transcripts with both ends pinned (zkhip.verify); -> list of failing labels"""
    from zkhip.verify import check_dhyperplonk_transcripts, dhyperplonk_anchors

    anchors = dhyperplonk_anchors(values_by_party, me, n_parties)
    want = len(values_by_party[0]) if leader else 7
    bad = list(check_dhyperplonk_transcripts(n, res_sc, chal, n_parties, leader, echo, anchors=anchors))
    if len(anchors) != want:
        bad.append(f"{len(anchors)} anchors instead of {want}")
    return bad


def self_check(n, res, digests, pk, pp, ctx, net, run_seed, world, data_parallel=False):
    """size-independent properties of a finished run (see tests/test_gpu_e2e_fullsize.py); "ok" or the failures"""
    import numpy as np

    from zkhip import dist_primitive as dp
    from zkhip.field import random_fr

    res_t, mine, closing = traced_run(n, pk, pp, ctx, net, run_seed, data_parallel)
    if world > 1:
        import torch.distributed as dist

        vals = [None] * world
        dist.all_gather_object(vals, mine)
    else:
        vals = [mine]
    bad = anchored_failures(n, res_t, pk, vals, net.party_id if world > 1 else 0, pp.n, net.is_leader, world == 1)
    if not closing:
        bad.append("closing row of a c_sumcheck_product differs from pss2ss of the folded last values")
    if digest(res_t) != digests[-1]:
        bad.append("the traced (anchored) run differs from the timed ones")
    if len(set(digests)) != 1:
        bad.append("repetitions disagree")
    (gate_proofs, gate_comms), (w_proofs, w_commits, w_opens) = res
    T, M = pk.tables, 1 << n
    hlen = 4 * M // pp.n
    same = lambda a, b: (np.asarray(a[0]) == np.asarray(b[0])).all() and np.asarray(a[1]).shape == np.asarray(b[1]).shape and (np.asarray(a[1]) == np.asarray(b[1])).all()
    # one call at a time (collective calls: every rank takes part)
    local_s_p = ctx.to_device(random_fr(hlen, run_seed * 31 + 1))
    if not (w_commits[0] == dp.d_commit(ctx, pk.d_commitment, local_s_p, hlen, net)).all():
        bad.append("d_commit(local_s) differs from the single call")
    if not same(w_opens[4], dp.d_open(ctx, pk.d_commitment, T["sid_p"], hlen, pk.challenge_r2, net)):
        bad.append("d_open(sid_p) differs from the single call")
    if not same(w_opens[1], dp.c_open(ctx, pk.c_commitment, T["V"], 4 * M, pk.challenge_r2, pp, net)):
        bad.append("c_open(V) differs from the single call")
    if not (gate_comms[0][0] == dp.c_commit(ctx, pk.c_commitment, [T["a_evals"]], [pk.lens["a_evals"]], pp, net)[0]).all():
        bad.append("c_commit(a) differs from the single call")
    return "ok" if not bad else bad


def party_threads(args):
    """the 8-party protocol with every party on its own thread and ctx, all on GPU 0: no echo shortcut anywhere"""
    import zkhip
    from zkhip.hyperplonk import PackedProvingParameters, dhyperplonk
    from zkhip.net import LocalTestNet
    from zkhip.pss import PackedSharingParams
    from types import SimpleNamespace

    pp = PackedSharingParams(1)

    def party(net):
        ctx = zkhip.Ctx(0)
        try:
            pk = PackedProvingParameters.new(args.n, pp, ctx, seed=0x5CA1AB1E % 1000 + net.party_id, chal_seed=0xC4A1, window_tables=not args.no_window_tables)
            best, digests, res = None, [], None
            for r in range(args.reps + 1):
                res, timers = dhyperplonk(args.n, pk, pp, ctx, net, seed=7 + net.party_id, data_parallel=args.data_parallel)
                digests.append(digest(res))
                if net.is_leader and (r > 0 or args.reps == 0) and (best is None or timers["Distributed HyperPlonk"] < best["Distributed HyperPlonk"]):
                    best = timers
            bad, chk = [], None
            if not args.no_check:
                res_t, values, closing = traced_run(args.n, pk, pp, ctx, net, 7 + net.party_id, args.data_parallel)
                if not closing:
                    bad.append("closing row of a c_sumcheck_product differs from pss2ss of the folded last values")
                if digest(res_t) != digests[-1]:
                    bad.append("the traced (anchored) run differs from the timed ones")
                chk = (((res_t[0][0], None), (res_t[1][0], None, None)), values,
                       SimpleNamespace(challenge=pk.challenge, challenge_r1=pk.challenge_r1, challenge_r2=pk.challenge_r2))
            if len(set(digests)) != 1:
                bad.append("repetitions disagree")
            return best, bad, (net.upload, net.download), chk
        finally:
            ctx.close()

    out = LocalTestNet.simulate_network_round(pp.n, party)
    bad = [f"party {p}: {b}" for p, (_, bs, _, _) in enumerate(out) for b in bs]
    if not args.no_check:  # chains anchored with ALL parties' values (the leader's d_ rows are sums over the parties)
        vals = [o[3][1] for o in out]
        for p, o in enumerate(out):
            bad += [f"party {p}: {b}" for b in anchored_failures(args.n, o[3][0], o[3][2], vals, p, pp.n, p == 0, False)]
    print(json.dumps({"n": args.n, "l": 1, "parties": pp.n, "mode": "8 party threads, one ctx each, ALL on GPU 0 (one GPU does the work of eight)", "timers_s": out[0][0],
                      "comm_bytes": list(out[0][2]), "checks": "ok" if not bad else bad, "check_kind": "skipped" if args.no_check else "anchored"}))
    if bad:
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", "--nvars", dest="n", type=int, default=14)  # use --nvars under torch.distributed.run (its own parser grabs --n)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--data-parallel", action="store_true")
    ap.add_argument("--net", choices=("rccl", "torch"), default="rccl", help="multi-rank exchanges: the C-ABI communicator (RCCL inside the ctx, device-resident) or torch.distributed")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-window-tables", action="store_true", help="MSMs without the precomputed SRS window tables (zk_srs_precompute)")
    ap.add_argument("--table-max-log2", type=int, default=24, help="build MSM window tables for SRS levels up to 2^k points (memory: ~14 x 96 B per point)")
    ap.add_argument("--no-dedup", action="store_true", help="compute identical MSM items of a step separately (the two opens of V share their first quotient's commitment by default)")
    ap.add_argument("--party-threads", action="store_true", help="all 8 parties as threads of this process, one ctx each on GPU 0 (real 8-party exchanges and point combinations, one GPU doing eight GPUs' work)")
    args = ap.parse_args()
    if args.no_dedup:
        from zkhip import dist_primitive as _dp

        _dp.DEDUP_MSM = False
    if args.party_threads:
        return party_threads(args)
    import zkhip
    from zkhip.hyperplonk import PackedProvingParameters, dhyperplonk
    from zkhip.pss import PackedSharingParams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    lrank = int(os.environ.get("LOCAL_RANK", "0"))
    pp = PackedSharingParams(1)
    if world > 1:
        import torch
        import torch.distributed as dist
        from zkhip.net import TorchDistNet

        # ZK_BENCH_BACKEND=gloo: exercise the 8-party exchanges on a box with fewer GPUs than ranks
        backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")
        lrank = lrank % torch.cuda.device_count()
        torch.cuda.set_device(lrank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", lrank))
        else:
            dist.init_process_group(backend)
        assert world == pp.n, "l = 1 needs exactly 8 parties"
    ctx = zkhip.Ctx(lrank)
    if world > 1:
        from zkhip.net import RcclNet

        if args.net == "rccl" and backend == "nccl":
            net = RcclNet.from_torch_dist(ctx)
        else:
            net = TorchDistNet(device=torch.device("cuda", lrank) if backend == "nccl" else None)
    else:
        from zkhip.net import LeaderEchoNet

        net = LeaderEchoNet(pp.n)
    t0 = time.perf_counter()
    pk = PackedProvingParameters.new(args.n, pp, ctx, seed=0x5CA1AB1E % 1000 + rank, chal_seed=0xC4A1, window_tables=not args.no_window_tables, table_max_log2=args.table_max_log2)  # challenges are shared public values
    setup = time.perf_counter() - t0
    best, digests, res = None, [], None
    for r in range(args.reps + 1):
        res, timers = dhyperplonk(args.n, pk, pp, ctx, net, seed=7 + rank, data_parallel=args.data_parallel)
        digests.append(digest(res))
        if r > 0 or args.reps == 0:
            if best is None or timers["Distributed HyperPlonk"] < best["Distributed HyperPlonk"]:
                best = timers
    checks = "skipped"
    if not args.no_check:
        checks = self_check(args.n, res, digests, pk, pp, ctx, net, 7 + rank, world, args.data_parallel)
    if rank == 0:
        print(json.dumps({"n": args.n, "l": 1, "parties": pp.n, "mode": ("comm(" + os.environ.get("ZK_BENCH_BACKEND", "nccl") + "," + type(net).__name__ + ")") if world > 1 else "leader-echo",
                          "setup_s": setup, "timers_s": best, "comm_bytes": [net.upload, net.download], "checks": checks, "check_kind": "skipped" if args.no_check else "anchored"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if checks not in ("ok", "skipped"):
        print("rank", rank, "self-check failed:", checks, file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    if os.environ.get("ZK_PROFILE") and int(os.environ.get("RANK", "0")) == 0:  # host hot spots of rank 0
        import cProfile, pstats

        cProfile.run("main()", "/tmp/zk_e2e.prof")
        pstats.Stats("/tmp/zk_e2e.prof").sort_stats("tottime").print_stats(25)
    else:
        main()
