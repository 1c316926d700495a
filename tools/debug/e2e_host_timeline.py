"""host-side timeline of one n-variable proof (leader mode): wall time between the calls of the protocol driver, by wrapping the dist_primitive /
backend entry points it uses.   python tools/debug/e2e_host_timeline.py <n>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import zkhip
from zkhip import dist_primitive as dp, hyperplonk as hp
from zkhip.net import LeaderEchoNet
from zkhip.pss import PackedSharingParams
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pp = PackedSharingParams(1)
ctx = zkhip.Ctx(0)
pk = hp.PackedProvingParameters.new(n, pp, ctx, seed=321, chal_seed=0xC4A1)
net = LeaderEchoNet(8)
for _ in range(2):
    hp.dhyperplonk(n, pk, pp, ctx, net, seed=7)
log, T0 = [], [0.0]
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); log.append((name, t - T0[0], time.perf_counter() - t)); return r
    setattr(obj, name, w)
for name in ("c_sumcheck_product", "c_sumcheck_product_many", "d_sumcheck_product_many", "c_open_many_q", "d_open_many_q", "d_commit_many_q", "c_commit_q", "d_acc_product", "open_many_q", "sumcheck_product", "pss2ss"):
    wrap(dp, name)
for name in ("fr_axpb", "fr_batch_div", "fr_deinterleave", "fr_add", "fr_sub", "to_device", "alloc", "copy_d2d", "msm_g1_batch_async", "g1_lincomb_batch", "sumcheck_batch"):
    wrap(ctx, name)
orig_start, orig_finish = dp.MsmQueue.start, dp.MsmQueue.finish
def st(self):
    t = time.perf_counter(); orig_start(self); log.append(("MsmQueue.start", t - T0[0], time.perf_counter() - t))
def fi(self):
    t = time.perf_counter(); r = orig_finish(self); log.append(("MsmQueue.finish", t - T0[0], time.perf_counter() - t)); return r
dp.MsmQueue.start, dp.MsmQueue.finish = st, fi
ctx.sync()
T0[0] = time.perf_counter()
res, tm = hp.dhyperplonk(n, pk, pp, ctx, net, seed=7)
total = time.perf_counter() - T0[0]
print("timers", {k: round(v * 1e3, 2) for k, v in tm.items()}, "wall incl. the inputs drawn before the timer", round(total * 1e3, 2))
prev_end = 0.0
for name, t, d in log:
    if d > 2e-4 or t - prev_end > 3e-4:
        print(f"  +{t*1e3:8.2f} ms  {name:26s} {d*1e3:8.3f} ms   (host gap before: {(t - prev_end)*1e3:6.3f} ms)")
    prev_end = max(prev_end, t + d)
