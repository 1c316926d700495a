"""the collaborative (packed) permutation check alone (hyperplonk/src/dhyperplonk.rs:1249-1385), leader-echo l = 1:
   python tools/cpermcheck_time.py [n = 20] [reps = 3]   -> one JSON line (best of reps)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.hyperplonk import PackedProvingParameters, cpermcheck
from zkhip.net import LeaderEchoNet
from zkhip.pss import PackedSharingParams
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pp = PackedSharingParams(1)
ctx = zkhip.Ctx(0)
pk = PackedProvingParameters.new(n, pp, ctx, seed=11)
net = LeaderEchoNet(pp.n)
best, t_acc = None, None
for r in range(reps + 1):
    ctx.sync()
    t0 = time.perf_counter()
    res, timers = cpermcheck(n, pk, pp, ctx, net, seed=5)
    dt = time.perf_counter() - t0
    if r and (best is None or dt < best):
        best = dt
# the masked product tree + share exchange alone (c_acc_product_and_share, dacc_product.rs:66-292)
from zkhip import dist_primitive as dp
T = pk.tables
G4 = 4 * ((1 << n) // pp.l)
num = ctx.fr_axpb(T["V"], T["sid"], pk.alpha, pk.beta, G4)
for r in range(reps + 1):
    ctx.sync()
    t0 = time.perf_counter()
    out = dp.c_acc_product_and_share(ctx, num, T["mask"], T["unmask0"], T["unmask1"], T["unmask2"], G4, pp, net)
    ctx.sync()
    dt = time.perf_counter() - t0
    if r and (t_acc is None or dt < t_acc):
        t_acc = dt
print(json.dumps({"n": n, "l": 1, "mode": "leader-echo", "cpermcheck_s": best, "c_acc_product_and_share_s": t_acc, "table_len": G4,
                  "proofs": len(res[0]), "commits": len(res[1]), "opens": len(res[2])}))
