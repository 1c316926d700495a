"""wall time per call of the sumcheck family through the C ABI (results come back to the host every call)
   python tools/sc_time.py [log2 sizes ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
only = os.environ.get("SC_MODE")  # one of product / plain / fold / open: counter-collection runs (tools/sc_pmc.sh)
sizes = [int(x) for x in sys.argv[1:]] or [12, 16, 20, 22, 24]
for lg in sizes:
    n = 1 << lg
    f, g, ch = ctx.to_device(random_fr(n, 1)), ctx.to_device(random_fr(n, 2)), random_fr(lg, 3)
    out, q = ctx.alloc(32), ctx.alloc(32 * n)
    for name, fn, byt in (("product", lambda: ctx.sumcheck_product(f, g, n, ch), 64), ("plain", lambda: ctx.sumcheck(f, n, ch), 32),
                          ("fold", lambda: (ctx.fold(f, n, ch, out=out), ctx.sync()), 32), ("open", lambda: ctx.open_rounds(f, n, ch, q_out=q), 64)):
        if only and name != only: continue
        for _ in range(3): fn()
        R = 20 if lg <= 22 else 5
        if only: print(f"calls {name} {3 + R}")
        t0 = time.perf_counter()
        for _ in range(R): fn()
        dt = (time.perf_counter() - t0) / R
        print(f"{name:8s} 2^{lg}: {dt*1e6:9.1f} us   {byt * n / dt / 1e9:8.1f} GB/s algorithmic", flush=True)
