import sys, time
sys.path[:0] = ["scalable-collaborative-zksnark_amd", "tests", "oracle"]
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
for n in (1500, (1 << 12) + 5, (1 << 16) + 63, (1 << 20) + 3, 1 << 22):
    srs = ctx.srs_generate(31, 77, n)
    sc = ctx.to_device(random_fr(n, 9))
    ref = ctx.msm_g1(srs, sc, n)
    for rec in (96, 128):
        out = {}
        for mode in (0, 1):
            ctx.dbg_tune("srs_table_batched", mode)
            ctx.sync(); t0 = time.perf_counter()
            srs.precompute(0, record_bytes=rec)
            ctx.sync(); dt = time.perf_counter() - t0
            out[mode] = (ctx.msm_g1(srs, sc, n), dt)
        assert (out[0][0] == ref).all() and (out[1][0] == ref).all(), (n, rec)
        print(f"n={n} rec={rec}: table build {out[0][1]*1e3:8.1f} ms (one inversion per record) -> {out[1][1]*1e3:8.1f} ms (batched); MSM results identical", flush=True)
    srs.free()
ctx.dbg_tune("srs_table_batched", 1)
