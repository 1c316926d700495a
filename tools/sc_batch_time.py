"""the layered product sumchecks of one wiring identity (3 per halving slice, hyperplonk/src/dhyperplonk.rs:417-478) one call at a time vs
   as ONE zk_sumcheck_batch:  python tools/sc_batch_time.py [log2 of the first slice = 18]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import random_fr
lg0 = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ctx = zkhip.Ctx(0)
n = 1 << (lg0 + 1)
tabs = [ctx.to_device(random_fr(n, 1 + i)) for i in range(4)]
reqs, off, clen = [], 0, 1 << lg0
while clen >= 1:
    ch = random_fr(24, 100 + clen.bit_length())
    at = lambda t: tabs[t].at(32 * off)
    reqs += [("product", at(0), at(1), clen, ch), ("product", at(0), at(2), clen, ch), ("product", at(2), at(3), clen, ch)]
    off += clen // 2 if clen > 1 else 0
    clen //= 2
def single():
    return [ctx.sumcheck_product(r[1], r[2], r[3], r[4]) for r in reqs]
def batch():
    return ctx.sumcheck_batch(reqs)
a, b = single(), batch()
assert all((x == y).all() for p, q in zip(a, b) for x, y in zip(p, q))
for name, fn in (("one call each", single), ("one batch", batch)):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R): fn()
    dt = (time.perf_counter() - t0) / R
    print(f"{len(reqs)} product sumchecks 2^{lg0} .. 1, {name}: {dt*1e3:.3f} ms", flush=True)
