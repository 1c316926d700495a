"""one MSM per size through the C ABI, default path vs precomputed SRS table (zk_srs_precompute): wall ms per call
   python tools/msm_time.py [log2 sizes ...]   (ZK_PRE_C=<bits> overrides the table's window width)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
for lg in [int(x) for x in sys.argv[1:]] or [14, 16, 18, 20, 22]:
    n = 1 << lg
    srs = ctx.srs_generate(123, 457, n)
    sc = ctx.to_device(random_fr(n, 5))
    R = 10 if lg <= 20 else 4
    def run():
        for _ in range(2): ctx.msm_g1(srs, sc, n)
        t0 = time.perf_counter()
        for _ in range(R): out = ctx.msm_g1(srs, sc, n)
        return (time.perf_counter() - t0) / R, out
    only_table = os.environ.get("ZK_ONLY_TABLE") == "1"  # (profiles of the window-table path alone)
    t_def, ref = (float("nan"), None) if only_table else run()
    srs.precompute(int(os.environ.get("ZK_PRE_C", "0")))
    t_pre, got = run()
    assert only_table or (got == ref).all()
    print(f"2^{lg}: default {t_def*1e3:7.3f} ms ({n/t_def/1e8:5.2f}e8/s)   precomputed {t_pre*1e3:7.3f} ms ({n/t_pre/1e8:5.2f}e8/s)   x{t_def/t_pre:.3f}", flush=True)
