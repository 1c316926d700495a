"""A/B of the MSM sort phase (round 6): level 1 straight from the scalars (msm_fused_min) and level 2 in LDS-staged tiles (msm_l2_tiled)
against the round-5 kernels, on window-table MSMs.   python tools/sort_ab.py [log2 sizes ...]
prints wall ms per call and ctx.msm_last_timing() = [digits+sort, accumulation, fix-up, reduction, host, total]; results must be bit-identical"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
MODES = [("r05", -1, -1), ("fused", 0, -1), ("tiled", -1, 0), ("both", 0, 0)]
for lg in [int(x) for x in sys.argv[1:]] or [16, 18, 20, 22, 24]:
    n = 1 << lg
    srs = ctx.srs_generate(123, 457, n)
    sc = ctx.to_device(random_fr(n, 5))
    srs.precompute(0)
    outs = {}
    for name, fused, tiled in MODES + MODES[:1] + MODES[-1:]:
        ctx.dbg_tune("msm_fused_min", fused)
        ctx.dbg_tune("msm_l2_tiled", tiled)
        for _ in range(3): ctx.msm_g1(srs, sc, n)
        R = 20 if lg <= 20 else 5
        t0 = time.perf_counter()
        for _ in range(R): out = ctx.msm_g1(srs, sc, n)
        t = (time.perf_counter() - t0) / R
        outs[name] = out
        print(f"2^{lg} {name:6s}: {t*1e3:8.3f} ms  {n/t/1e8:.3f}e8/s  phases {[round(float(x), 3) for x in ctx.msm_last_timing()]}", flush=True)
    for k in outs:
        assert (outs[k] == outs["r05"]).all(), k
    srs.free()
print("all modes bit-identical")
