#!/usr/bin/env python3
"""
End-to-end collaborative HyperPlonk (hyperplonk/src/dhyperplonk.rs) on MI355X.

  python tools/hyperplonk_bench.py --n 16                       # `leader` mode: party 0 alone, no-comm echo net (config 1 style)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/hyperplonk_bench.py --nvars 20   # l = 1, 8 parties = 8 GPUs

Prints the reference's timer sections (Commit / Gate identity / Wire identity / Open / total)
and the Comm: (up, down) byte counters for the leader.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", "--nvars", dest="n", type=int, default=14)  # use --nvars under torch.distributed.run (its own parser grabs --n)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--data-parallel", action="store_true")
    args = ap.parse_args()
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
        net = TorchDistNet(device=torch.device("cuda", lrank) if backend == "nccl" else None)
    else:
        from zkhip.net import LeaderEchoNet

        net = LeaderEchoNet(pp.n)
    ctx = zkhip.Ctx(lrank)
    t0 = time.perf_counter()
    pk = PackedProvingParameters.new(args.n, pp, ctx, seed=0x5CA1AB1E % 1000 + rank)
    setup = time.perf_counter() - t0
    best = None
    for r in range(args.reps + 1):
        res, timers = dhyperplonk(args.n, pk, pp, ctx, net, seed=7 + rank, data_parallel=args.data_parallel)
        if r > 0 or args.reps == 0:
            if best is None or timers["Distributed HyperPlonk"] < best["Distributed HyperPlonk"]:
                best = timers
    if rank == 0:
        print(json.dumps({"n": args.n, "l": 1, "parties": pp.n, "mode": ("comm(" + os.environ.get("ZK_BENCH_BACKEND", "nccl") + ")") if world > 1 else "leader-echo",
                          "setup_s": setup, "timers_s": best, "comm_bytes": [net.upload, net.download]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if os.environ.get("ZK_PROFILE") and int(os.environ.get("RANK", "0")) == 0:  # host hot spots of rank 0
        import cProfile, pstats

        cProfile.run("main()", "/tmp/zk_e2e.prof")
        pstats.Stats("/tmp/zk_e2e.prof").sort_stats("tottime").print_stats(25)
    else:
        main()
