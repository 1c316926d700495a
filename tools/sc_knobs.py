import os, sys, time
sys.path.insert(0, "scalable-collaborative-zksnark_amd")
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
n = 1 << 20
f, g, ch = ctx.to_device(random_fr(n, 1)), ctx.to_device(random_fr(n, 2)), random_fr(20, 3)
def run(tag):
    ctx.dbg_tune("sc_ts", 3)
    ts = []
    for _ in range(12):
        ctx.sumcheck_product(f, g, n, ch); ts.append(ctx.sumcheck_last_timing().copy())
    ctx.dbg_tune("sc_ts", 0)
    for _ in range(5): ctx.sumcheck_product(f, g, n, ch)
    t0 = time.perf_counter()
    for _ in range(50): ctx.sumcheck_product(f, g, n, ch)
    w = (time.perf_counter() - t0) / 50
    m = np.median(np.array(ts[2:]), axis=0)
    print(f"{tag:28s} first stage {m[0]*1e3:6.1f} us  all launches {m[1]*1e3:6.1f} us  wall {w*1e6:6.1f} us", flush=True)
run("default")
for wg in (1, 3, 4, 8):
    ctx.dbg_tune("sc_pass_wg", wg); run(f"sc_pass_wg={wg}")
ctx.dbg_tune("sc_pass_wg", 0)
ctx.dbg_tune("sc_kp", 1); run("sc_kp=1"); ctx.dbg_tune("sc_kp", 2)
ctx.dbg_tune("sc_local_g", 128); run("sc_local_g=128"); ctx.dbg_tune("sc_local_g", 512); run("sc_local_g=512"); ctx.dbg_tune("sc_local_g", 256)
ctx.dbg_tune("sc_pre", 0); run("sc_pre=0"); ctx.dbg_tune("sc_pre", 1)
