"""how long do zk_msm_g1_batch_async (enqueue) and zk_msm_wait take on the host while other jobs are in flight?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import zkhip
from zkhip.field import random_fr
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << lg
ctx = zkhip.Ctx(0)
srs = ctx.srs_generate(12345, 6789, n)
srs.precompute(0)
d = [ctx.to_device(random_fr(n, 5 + i)) for i in range(2)]
for _ in range(3):
    ctx.msm_g1(srs, d[0], n)
jobs, enq, wt = [], [], []
ctx.sync()
T0 = time.perf_counter()
for i in range(24):
    t = time.perf_counter(); jobs.append(ctx.msm_g1_batch_async([srs], [d[i & 1]], [n])); enq.append(time.perf_counter() - t)
    if len(jobs) >= depth:
        t = time.perf_counter(); jobs.pop(0).wait(); wt.append(time.perf_counter() - t)
for j in jobs:
    t = time.perf_counter(); j.wait(); wt.append(time.perf_counter() - t)
tot = time.perf_counter() - T0
print(f"depth {depth}: {tot / 24 * 1e3:.3f} ms per MSM; enqueue ms:", " ".join(f"{x*1e3:.2f}" for x in enq[:12]), "| wait ms:", " ".join(f"{x*1e3:.2f}" for x in wt[:12]), flush=True)
