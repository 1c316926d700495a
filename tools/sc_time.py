import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
for lg in (12, 16, 20, 22):
    n = 1 << lg
    f, g, ch = ctx.to_device(random_fr(n, 1)), ctx.to_device(random_fr(n, 2)), random_fr(lg, 3)
    for name, fn in (("product", lambda: ctx.sumcheck_product(f, g, n, ch)), ("plain", lambda: ctx.sumcheck(f, n, ch))):
        for _ in range(3): fn()
        t0 = time.perf_counter(); R = 20
        for _ in range(R): fn()
        print(f"sumcheck {name} 2^{lg}: {(time.perf_counter()-t0)/R*1e6:8.1f} us", flush=True)
