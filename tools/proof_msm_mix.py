"""the MSM batch of the wiring identity of an n-constraint proof (hyperplonk/src/dhyperplonk.rs:262-514, leader mode l = 1) as ONE
   zk_msm_g1_batch: python tools/proof_msm_mix.py [n = 20] [reps = 5]      (ZKHIP_TUNE=msm_serial=1: per-kernel times = work)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = zkhip.Ctx(0)
geo = lambda m: [1 << k for k in range(m, -1, -1)]
sizes = 2 * geo(n + 1) + 9 * [1 << (n - 1)] + 6 * geo(n - 2)
for k in range(n - 3, 0, -1):
    sizes += 3 * geo(k)
levels = {}
for s in sorted(set(sizes)):
    lv = ctx.srs_generate(100 + s, 457, s)
    if 64 <= s <= (1 << 22):
        lv.precompute(0)
    levels[s] = (lv, ctx.to_device(random_fr(s, s % 977)))
srs, sc = [levels[s][0] for s in sizes], [levels[s][1] for s in sizes]
for _ in range(2):
    ctx.msm_g1_batch(srs, sc, sizes)
t0 = time.perf_counter()
for _ in range(reps):
    ctx.msm_g1_batch(srs, sc, sizes)
dt = (time.perf_counter() - t0) / reps
print(f"wiring batch n={n}: {len(sizes)} items, {sum(sizes)} points: {dt*1e3:.2f} ms  ({sum(sizes)/dt:.3e} pts/s)")
