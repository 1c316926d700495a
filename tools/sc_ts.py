"""ZKHIP_TUNE=sc_ts=1 python tools/sc_ts.py <mode> <log2 size>: the stage timestamps of the local launches (stderr of the library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import zkhip
from zkhip.field import random_fr
mode, lg = sys.argv[1], int(sys.argv[2])
n = 1 << lg
ctx = zkhip.Ctx(0)
f, g, ch = ctx.to_device(random_fr(n, 1)), ctx.to_device(random_fr(n, 2)), random_fr(lg, 3)
for _ in range(4):
    if mode == "product": ctx.sumcheck_product(f, g, n, ch)
    elif mode == "plain": ctx.sumcheck(f, n, ch)
    elif mode == "open": ctx.open_rounds(f, n, ch)
    else: ctx.fold(f, n, ch); ctx.sync()
