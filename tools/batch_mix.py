"""time msm_g1_batch on a mix of item sizes: python tools/batch_mix.py 21,21  |  21,20,19,21,20,19  |  19x11,18x10 ...
   BATCH_TABLE=1: every level with its window table (zk_srs_precompute)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
levels = {}
for spec in sys.argv[1:]:
    lgs = []
    for x in spec.split(","):
        lg, _, rep = x.partition("x")
        lgs += [int(lg)] * int(rep or 1)
    for lg in lgs:
        if lg not in levels:
            levels[lg] = (ctx.srs_generate(100 + lg, 457, 1 << lg), ctx.to_device(random_fr(1 << lg, lg)))
            if os.environ.get("BATCH_TABLE") and lg >= 6:
                levels[lg][0].precompute(0)
    srs, sc, ns = [levels[lg][0] for lg in lgs], [levels[lg][1] for lg in lgs], [1 << lg for lg in lgs]
    for _ in range(2): ctx.msm_g1_batch(srs, sc, ns)
    R = 5; ph = np.zeros(6); t0 = time.perf_counter()
    for _ in range(R):
        ctx.msm_g1_batch(srs, sc, ns); ph += ctx.msm_last_timing()
    dt = (time.perf_counter() - t0) / R
    print(f"{spec:28s} {dt*1e3:8.3f} ms  first class: sort={ph[0]/R:.3f} acc={ph[1]/R:.3f} fix={ph[2]/R:.3f} red={ph[3]/R:.3f} host={ph[4]/R:.3f}  ({sum(ns)/dt:.3e} pts/s)", flush=True)
