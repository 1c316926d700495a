"""aggregate MSM throughput of N host threads, a ctx each, all on GPU 0: does the hardware overlap one MSM's latency-bound
tail with another's accumulation?   python tools/debug/two_ctx_msm.py <log2 n> <threads> [reps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import zkhip
from zkhip.field import random_fr
lg, nt = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 1 << lg
ctxs = [zkhip.Ctx(0) for _ in range(nt)]
srs = [c.srs_generate(0x1234567 + i, 0x89ABCDE + 7 * i, n) for i, c in enumerate(ctxs)]
for s in srs:
    s.precompute(0)
sc = [c.to_device(random_fr(n, 5 + i)) for i, c in enumerate(ctxs)]
bar = threading.Barrier(nt + 1)
def work(i):
    for _ in range(3):
        ctxs[i].msm_g1(srs[i], sc[i], n)
    bar.wait()
    for _ in range(reps):
        ctxs[i].msm_g1(srs[i], sc[i], n)
    bar.wait()
ths = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
for t in ths: t.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
for t in ths: t.join()
print(f"2^{lg} x {nt} threads: {dt / (reps * nt) * 1e3:.3f} ms per MSM aggregate ({reps * nt * n / dt:.3e} scalar-muls/s)", flush=True)
