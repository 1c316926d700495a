"""precomputed-SRS MSM timing sweep; run on the GPU box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np
import zkhip
from zkhip.field import random_fr
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
ctx = zkhip.Ctx(0)
srs = ctx.srs_generate(123, 457, n)
sc = ctx.to_device(random_fr(n, 5))
ref = ctx.msm_g1(srs, sc, n)
for c in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,17,18,19,20").split(",")]:
    t0 = time.perf_counter(); srs.precompute(c); tp = time.perf_counter() - t0
    got = ctx.msm_g1(srs, sc, n)
    assert (got == ref).all(), c
    ph = np.zeros(6); R = 5; t0 = time.perf_counter()
    for _ in range(R):
        ctx.msm_g1(srs, sc, n); ph += ctx.msm_last_timing()
    dt = (time.perf_counter() - t0) / R
    print(f"  log2n={log2n} shared c={c}: {dt*1e3:7.3f} ms  sort={ph[0]/R:.3f} acc={ph[1]/R:.3f} fix={ph[2]/R:.3f} red={ph[3]/R:.3f} host={ph[4]/R:.3f}  (precompute {tp*1e3:.0f} ms)", flush=True)
