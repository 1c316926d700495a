"""knob sweep of the round-6 sort on window-table MSMs: python tools/sort_ab2.py LOG2 key=value[,key=value] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
lg = int(sys.argv[1]); n = 1 << lg
srs = ctx.srs_generate(123, 457, n); sc = ctx.to_device(random_fr(n, 5)); srs.precompute(0)
ref = None
for spec in sys.argv[2:] * 2:
    sets = [kv.split("=") for kv in spec.split(",") if kv]
    for k, v in sets: ctx.dbg_tune(k, int(v))
    for _ in range(3): ctx.msm_g1(srs, sc, n)
    R = 20 if lg <= 20 else 5
    t0 = time.perf_counter()
    for _ in range(R): out = ctx.msm_g1(srs, sc, n)
    t = (time.perf_counter() - t0) / R
    ref = out if ref is None else ref
    assert (out == ref).all()
    print(f"2^{lg} {spec:40s}: {t*1e3:8.3f} ms  phases {[round(float(x), 3) for x in ctx.msm_last_timing()]}", flush=True)
    for k, v in sets: ctx.dbg_tune(k, 0)
