"""blocking vs pipelined MSMs of one size: python tools/msm_pipeline.py <log2 n> [reps] [table 0/1]
   blocking: zk_msm_g1 back to back; pipelined: zk_msm_g1_batch_async with two jobs in flight (the tail of one job overlaps
   with the sort + accumulation of the next)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("scalable-collaborative-zksnark_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, zkhip
from zkhip.field import random_fr
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
table = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 1 << lg
ctx = zkhip.Ctx(0)
srs = ctx.srs_generate(12345, 6789, n)
if table:
    srs.precompute(0)
d = [ctx.to_device(random_fr(n, 5 + i)) for i in range(2)]
ref = [ctx.msm_g1(srs, d[i], n) for i in range(2)]
for _ in range(3):
    ctx.msm_g1(srs, d[0], n)
t = time.perf_counter()
for i in range(reps):
    ctx.msm_g1(srs, d[i & 1], n)
blocking = (time.perf_counter() - t) / reps
ph = ctx.msm_last_timing()
for depth in (2, 3):
    jobs = []
    ctx.sync()
    t = time.perf_counter()
    for i in range(reps):
        jobs.append((i, ctx.msm_g1_batch_async([srs], [d[i & 1]], [n])))
        if len(jobs) >= depth:
            k, j = jobs.pop(0)
            assert (j.wait()[0] == ref[k & 1]).all()
    for k, j in jobs:
        assert (j.wait()[0] == ref[k & 1]).all()
    piped = (time.perf_counter() - t) / reps
    print(f"2^{lg} table={table}: blocking {blocking*1e3:.3f} ms/MSM ({n/blocking:.3e}/s; phases sort {ph[0]:.3f} acc {ph[1]:.3f} fix {ph[2]:.3f} red {ph[3]:.3f} host {ph[4]:.3f})"
          f"  pipelined depth {depth}: {piped*1e3:.3f} ms/MSM ({n/piped:.3e}/s)", flush=True)
