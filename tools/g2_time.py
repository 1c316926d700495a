"""python tools/g2_time.py <lg> [table window bits ...]: time zk_msm_g2 on 2^lg points, without and with the window table (bases = an arithmetic sequence built with the oracle-free host arithmetic of zkhip.pairing)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip import pairing as pr
from zkhip.field import fq_mont, random_fr
ctx = zkhip.Ctx(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << lg
step, cur, rows = pr.g2_mul(pr.G2_GEN, 991), pr.g2_mul(pr.G2_GEN, 77), []
for _ in range(n):
    rows.append(np.concatenate([fq_mont(cur[0][0]), fq_mont(cur[0][1]), fq_mont(cur[1][0]), fq_mont(cur[1][1])]))
    cur = pr.g2_add(cur, step)
srs = ctx.srs_register_g2(np.array(rows, dtype=np.uint64))
sc = ctx.to_device(random_fr(n, 5))
def run(tag):
    for _ in range(2): out = ctx.msm_g2(srs, sc, n)
    t0 = time.perf_counter(); R = 5
    for _ in range(R): ctx.msm_g2(srs, sc, n)
    dt = (time.perf_counter() - t0) / R
    print(f"G2 MSM 2^{lg} {tag}: {dt*1e3:.2f} ms  {n/dt:.3e} scalar-muls/s  phases(ms) {[round(float(x),2) for x in ctx.msm_last_timing()]}", flush=True)
    return out
ref = run("no table")
for c in [int(x) for x in sys.argv[2:]] or [0]:  # window bits of the table (0: the library's pick)
    t0 = time.perf_counter()
    srs.precompute(c)
    tb = time.perf_counter() - t0
    assert (run(f"window table c={srs.table_window} (built in {tb:.2f} s)") == ref).all()
