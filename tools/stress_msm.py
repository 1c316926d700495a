"""randomised differential test of the MSM against the C oracle: python tools/stress_msm.py [seconds]
   STRESS_TABLE=1: on an SRS with a window table (zk_srs_precompute), rebuilt with a random window width every 25 cases"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("scalable-collaborative-zksnark_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, zkhip, coracle as co
from helpers import jac_norm_to_affine, rand_fr, synthetic_bases
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "12345")))  # STRESS_SEED: another sequence of cases
ctx = zkhip.Ctx(0)
NMAX = 1 << 16
bases, _ = synthetic_bases(NMAX, 4242)
srs = ctx.srs_register(bases)
table = bool(os.environ.get("STRESS_TABLE"))
t0 = time.time(); it = 0; bad = 0; tc = None
while time.time() - t0 < budget:
    it += 1
    if table and it % 25 == 1:
        tc = int(rng.choice([0, 4, 7, 10, 13, 16, 17, 18, 19, 20]))
        srs.precompute(tc)
    mode = rng.integers(0, 4)
    if mode == 0:   # single MSM, random size / offset / window
        n = int(rng.integers(1, NMAX + 1)) if rng.random() < 0.5 else int(rng.integers(1, 600))
        off = int(rng.integers(0, NMAX - n + 1))
        c = int(rng.choice([0, 0, 0, 2, 3, 5, 6, 9, 10, 12, 14, 17, 18, 20]))
        sc = rand_fr(n, 1000 + it)
        if rng.random() < 0.3:  # skew: few distinct scalars
            sc = sc[rng.integers(0, min(n, 3), size=n)]
        ctx.msm_set_window(c)
        got = jac_norm_to_affine(ctx.msm_g1(srs, ctx.to_device(sc), n, offset=off))
        ctx.msm_set_window(0)
        exp = co.msm_g1(bases[off:off + n], sc)
        ok = (got == exp).all()
        desc = f"single n={n} off={off} c={c} table={tc}"
    else:           # batch of mixed sizes (window classes, host pool, completion order)
        k = int(rng.integers(2, 12))
        ns = [int(2 ** rng.uniform(0, 15.5)) for _ in range(k)]
        offs = [int(rng.integers(0, NMAX - n + 1)) for n in ns]
        scs = [rand_fr(n, 5000 + 50 * it + j) for j, n in enumerate(ns)]
        got = ctx.msm_g1_batch([srs] * k, [ctx.to_device(s) for s in scs], ns, offsets=offs)
        ok = all((jac_norm_to_affine(got[j]) == co.msm_g1(bases[offs[j]:offs[j] + ns[j]], scs[j])).all() for j in range(k))
        desc = f"batch ns={ns} table={tc}"
    if not ok:
        bad += 1
        print("MISMATCH", desc, flush=True)
print(f"stress: {it} cases in {time.time() - t0:.0f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
