"""time one msm_g1_batch of k items of 2^lg points: python tools/batch_time.py <lg> <k> [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
lg, k = int(sys.argv[1]), int(sys.argv[2]); R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 1 << lg
ctx = zkhip.Ctx(0)
srs = ctx.srs_generate(123, 457, n)
sc = [ctx.to_device(random_fr(n, 5 + i)) for i in range(k)]
for _ in range(2): ctx.msm_g1_batch([srs] * k, sc, [n] * k)
ph = np.zeros(6); t0 = time.perf_counter()
for _ in range(R):
    ctx.msm_g1_batch([srs] * k, sc, [n] * k); ph += ctx.msm_last_timing()
dt = (time.perf_counter() - t0) / R
print(f"batch {k} x 2^{lg}: {dt*1e3:7.3f} ms  sort={ph[0]/R:.3f} acc={ph[1]/R:.3f} fix={ph[2]/R:.3f} red={ph[3]/R:.3f} host={ph[4]/R:.3f}  ({k*n/dt:.3e} pts/s)")
