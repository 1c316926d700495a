"""k_fixup_long in a batch of equal-size window-table items: python tools/debug/batch_long.py <log2 n> <items>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import zkhip
from zkhip.field import random_fr
lg, k = int(sys.argv[1]), int(sys.argv[2])
n = 1 << lg
ctx = zkhip.Ctx(0)
srs = ctx.srs_generate(12345, 6789, n)
srs.precompute(0)
sc = [ctx.to_device(random_fr(n, 5 + i)) for i in range(k)]
for _ in range(3):
    out = ctx.msm_g1_batch([srs] * k, sc, [n] * k)
t0 = time.perf_counter()
for _ in range(5):
    out = ctx.msm_g1_batch([srs] * k, sc, [n] * k)
dt = (time.perf_counter() - t0) / 5
print(f"{k} x 2^{lg}: {dt*1e3:.3f} ms per batch, {k*n/dt:.3e} scalar-muls/s, phases {[round(float(x),3) for x in ctx.msm_last_timing()]}", flush=True)
