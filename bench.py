#!/usr/bin/env python3
"""
bench.py -- the reference's headline metric on MI355X: d_msm G1 scalar-muls/s (+ d_sumcheck Fr
field-ops/s) at 2^20 shares (BASELINE.json configs[1]/[2]).

A "step" = one party's d_msm on 2^20 packed BLS12-381 G1 shares: the local `G::msm`
(dist-primitive/src/dmsm.rs:19-24) on HBM-resident bases and scalars, followed -- when more
than one rank runs -- by the d_msm exchange (dmsm.rs:29-40) as one RCCL all-gather of the
144-byte results plus the replicated public linear map (at 8 ranks this is exactly the l = 1,
8-party d_msm).  One process per GPU, rank = party: per-GPU work is fixed => "scaling": "weak".

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = bucket accumulation, HIP-event
timed inside the library on its own stream) and `cpu_baseline` (oracle = C port of the
reference's single-threaded ark-ec Pippenger, timed on the host cores of this box).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))

import numpy as np  # noqa: E402

LOG2_N = 20
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FQ_MUL_PEAK = 80.0e9  # measured Fq multiplier rate (13x30-bit limbs, csrc/fq30.cuh) at the kernel's 2 waves/SIMD, whole chip (profiles/r01_ubench_mul30.txt)
FR_MUL_PEAK = 133.0e9  # measured Fr Montgomery mul/s


def pmc_traffic_bytes(kernel: str):
    """
    Memory-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/*pmc_hbm_traffic.csv: FETCH_SIZE and WRITE_SIZE, KB per dispatch, separate passes).
    FETCH_SIZE is doubled, as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B):
    tools/ubench_gather.hip confirms the factor for this kernel's own pattern -- random 96-byte records are
    1.5 lines of 128 B on average, the counter reports 96 B per record (profiles/r01_fetch_calibration.txt).
    WRITE_SIZE is taken raw (uncalibrated, 6 % of the total).  Infinity-Cache hits are included.
    """
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.csv")))
    if not files:
        return None
    tot = 0.0
    for row in csv.reader(open(files[-1])):
        if len(row) == 4 and row[1] == kernel:
            tot += float(row[3]) * 1024.0 * (2.0 if row[0] == "FETCH_SIZE" else 1.0)
    return tot or None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=LOG2_N)
    ap.add_argument("--cpu-log2n", type=int, default=20, help="size of the bounded CPU-baseline MSM sample")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch

    import zkhip
    from zkhip.field import random_fr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # ZK_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than
    # ranks (ranks then share GPUs round-robin); the real runs use nccl (= RCCL over xGMI).
    backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = 1 << args.log2n
    ctx = zkhip.Ctx(gpu)  # raises if libzkhip.so / the GPU is missing (no fallback)

    # ---- synthetic inputs, resident in HBM before the timed region ----
    seed = 0x5CA1AB1E + 1000 * 2 + rank
    srs = ctx.srs_generate(0x1234567 + rank, 0x89ABCDE + 7 * rank, n)  # P_i = (k0 + i k1) G
    scal_np = random_fr(n, seed)
    scalars = torch.from_numpy(scal_np.view(np.int64)).to(dev)
    f_t = torch.from_numpy(random_fr(n, seed + 1).view(np.int64)).to(dev)
    g_t = torch.from_numpy(random_fr(n, seed + 2).view(np.int64)).to(dev)
    chal = random_fr(args.log2n, seed + 3)
    torch.cuda.synchronize()

    net = pp = None
    if world > 1:
        from zkhip.net import TorchDistNet
        from zkhip.pss import PackedSharingParams

        net = TorchDistNet(device=dev if backend == "nccl" else None)
        pp = PackedSharingParams(1) if world == 8 else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if world == 1:
            return ctx.msm_g1(srs, scalars, n)
        if pp is not None:  # the full 8-party d_msm
            from zkhip.dist_primitive import d_msm

            return d_msm(ctx, [srs], [scalars], [n], pp, net)
        local = ctx.msm_g1(srs, scalars, n)
        net.all_gather(local)  # partial party set: the exchange only
        return local

    for _ in range(args.warmup):
        step()
    phase = np.zeros(6)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        phase += ctx.msm_last_timing()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    phase /= max(args.steps, 1)

    # ---- secondary metric: d_sumcheck_product Phase-1 at 2^20 (dsumcheck.rs:377-429) ----
    for _ in range(2):
        ctx.sumcheck_product(f_t, g_t, n, chal)
    barrier()
    s0 = time.perf_counter()
    sc_reps = max(args.steps, 1)
    for _ in range(sc_reps):
        ctx.sumcheck_product(f_t, g_t, n, chal)
    barrier()
    sc_dt = (time.perf_counter() - s0) / sc_reps
    for _ in range(2):
        ctx.sumcheck(f_t, n, chal)
    barrier()
    s0 = time.perf_counter()
    for _ in range(sc_reps):
        ctx.sumcheck(f_t, n, chal)
    barrier()
    scp_dt = (time.perf_counter() - s0) / sc_reps

    # second figure: the same step with the scalars coming from host memory (PCIe-inclusive);
    # never `value`
    h2d_ms = None
    if world == 1:
        tmp = ctx.alloc(32 * n)
        for _ in range(2):
            tmp.upload(scal_np)
            ctx.msm_g1(srs, tmp, n)
        barrier()
        h0 = time.perf_counter()
        for _ in range(3):
            tmp.upload(scal_np)
            ctx.msm_g1(srs, tmp, n)
        barrier()
        h2d_ms = (time.perf_counter() - h0) / 3 * 1e3

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = world * n * args.steps / dt
    accum_ms = float(phase[1])
    alg_bytes = 128.0 * n  # 96-B affine point + 32-B scalar per scalar-mul, read once
    achieved = alg_bytes / (accum_ms * 1e-3) / 1e9 if accum_ms > 0 else 0.0
    c = ctx.lib.zk_msm_window(n)
    windows = (129 + c - 1) // c  # scalars are split into two 128-bit halves (k = k1 + k2*lambda): 2n entries per window
    fq_mul_equiv = 2.0 * n * windows * 10.0  # one XYZZ mixed add (8M + 2S) per entry per window
    out = {
        "metric": "G1 scalar-muls/sec (d_msm) + Fr field-ops/sec (d_sumcheck), 2^20 shares, 1/2/4/8 GPU",
        "value": value,
        "unit": "G1 scalar-muls/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"d_msm on 2^{args.log2n} packed BLS12-381 G1 shares per party (BASELINE.json configs[1]), l=1",
            "points_per_party": n,
            "parties": world,
            "exchange": "none" if world == 1 else ("d_msm all-gather + PSS unpack2/pack map" if world == 8 else "all-gather only (partial party set)"),
            "pippenger_window_bits": c,
            "windows": windows,
            "entries_per_window": 2 * n,
        },
        "ms_per_step_scalars_from_host": h2d_ms,
        "msm_phase_ms": {"digits_sort": float(phase[0]), "k_accum_tiles": accum_ms, "fixup": float(phase[2]), "bucket_reduce": float(phase[3]), "host_combine": float(phase[4])},
        "roofline": {
            "kernel": "zk::k_accum_tiles (bucket accumulation)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic_bytes("zk::k_accum_tiles") if args.log2n == 20 else None,
            "note": "integer-VALU bound, not HBM bound: see int_alu",
            "int_alu": {
                "achieved_fq_mul_per_s": fq_mul_equiv / (accum_ms * 1e-3) if accum_ms > 0 else 0.0,
                "measured_peak_fq_mul_per_s": FQ_MUL_PEAK,
                "frac": (fq_mul_equiv / (accum_ms * 1e-3) / FQ_MUL_PEAK) if accum_ms > 0 else 0.0,
            },
        },
        "sumcheck": {
            "workload": f"d_sumcheck_product phase 1, 2^{args.log2n} Fr shares x 2 tables, {args.log2n} rounds",
            "ms": sc_dt * 1e3,
            "fr_field_ops_per_s": 18.0 * n / sc_dt,  # reference op count: 9N mul + 9N add
            "fr_mul_as_written_per_s": 9.0 * n / sc_dt,
            "hbm_algorithmic_GBps": 64.0 * n / sc_dt / 1e9,
            "hbm_frac": 64.0 * n / sc_dt / 1e9 / HBM_PEAK_GBS,
            "plain": {  # d_sumcheck phase 1 (dsumcheck.rs:301-315): 2N mul + 3N add as the reference writes it
                "ms": scp_dt * 1e3,
                "fr_field_ops_per_s": 5.0 * n / scp_dt,
                "hbm_algorithmic_GBps": 32.0 * n / scp_dt / 1e9,
            },
        },
    }

    if not args.no_cpu and world == 1:  # the CPU leg runs at N = 1 only (rank 0)
        # CPU baseline leg: the oracle (C port of the reference's single-threaded path) on this
        # box's host cores.  Checker/baseline only -- never part of the measured GPU path.
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import coracle as co

        m = 1 << args.cpu_log2n
        bases_h = srs.download()[:m].copy()
        sc_h = scal_np[:m].copy()
        t = time.perf_counter()
        ref = co.msm_g1(bases_h, sc_h)
        cpu_dt = time.perf_counter() - t
        got = ctx.msm_g1(srs, ctx.to_device(sc_h), m)
        assert (got[:12] == ref).all(), "GPU MSM differs from the CPU oracle on the baseline sample"
        t = time.perf_counter()
        co.sumcheck_product(np.ascontiguousarray(f_t.cpu().numpy().view(np.uint64)[: 1 << 18]), np.ascontiguousarray(g_t.cpu().numpy().view(np.uint64)[: 1 << 18]), chal[:18])
        cpu_sc = time.perf_counter() - t
        # generous baseline: the same port on many host cores (the reference itself is single-threaded per
        # party: no `parallel` feature, Cargo.lock:120-134) -- contiguous chunks of the MSM on a thread
        # pool (ctypes releases the GIL), partial results added with the oracle's group law
        from concurrent.futures import ThreadPoolExecutor

        cores = max(1, min(64, (os.cpu_count() or 1) // 2))
        bounds = [m * i // cores for i in range(cores + 1)]
        t = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            parts = list(ex.map(lambda i: co.msm_g1(bases_h[bounds[i] : bounds[i + 1]], sc_h[bounds[i] : bounds[i + 1]]), range(cores)))
        import pyoracle as po
        from zkhip.field import affine_mont_to_ints

        acc = None
        for pt in parts:
            acc = po.g1_add(acc, affine_mont_to_ints(pt))
        cpu_mt = time.perf_counter() - t
        assert acc == affine_mont_to_ints(ref), "chunked CPU MSM differs from the single-thread result"
        out["cpu_baseline"] = {
            "value": m / cpu_dt,
            "unit": "G1 scalar-muls/s",
            "cores": 1,
            "kind": "port",
            "sample": f"one MSM of 2^{args.cpu_log2n} of the same bases/scalars (ark-ec window rule c={co.msm_window(m)}), {cpu_dt:.1f} s; result bit-identical to the GPU",
            "sumcheck_fr_field_ops_per_s": 18.0 * (1 << 18) / cpu_sc,
            "sumcheck_sample": f"sumcheck_product on 2^18 of the same tables, {cpu_sc:.2f} s",
            "all_cores": {"value": m / cpu_mt, "unit": "G1 scalar-muls/s", "cores": cores, "sample": f"same MSM in {cores} chunks on {cores} threads, {cpu_mt:.2f} s"},
            "host": os.uname().nodename,
            "nproc": os.cpu_count(),
        }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
