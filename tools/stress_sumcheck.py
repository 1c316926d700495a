"""randomised differential test of the sumcheck family against the C oracle: python tools/stress_sumcheck.py [seconds]
   random table lengths 2^0 .. 2^21 (weighted towards the stage boundaries 2^9 .. 2^20), all four modes, partial folds.
   `run(ctx, co, cases=..)` is the bounded form tests/test_gpu_bigsizes.py puts in the -m gpu suite."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("scalable-collaborative-zksnark_amd", "oracle", "tests"):
    if os.path.join(ROOT, p) not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np


def run(ctx, co, cases=None, seconds=None, seed=777, verbose=True):
    """-> number of mismatching cases; stops after `cases` cases or `seconds` seconds, whichever is given / comes first"""
    from helpers import rand_fr

    rng = np.random.default_rng(seed)
    t0 = time.time(); it = 0; bad = 0
    while (cases is None or it < cases) and (seconds is None or time.time() - t0 < seconds):
        it += 1
        lg = int(rng.choice(list(range(0, 22)) + [9, 10, 11, 12, 16, 17, 18, 18, 19, 19, 20]))
        n = 1 << lg
        mode = ("plain", "product", "fold", "open")[int(rng.integers(0, 4))]
        f = rand_fr(n, 10_000 + it)
        if rng.random() < 0.15:  # degenerate tables: zeros / one repeated value / r - 1
            f[:] = 0 if rng.random() < 0.5 else f[0]
        ch = rand_fr(max(lg, 1), 20_000 + it)
        d_f = ctx.to_device(f)
        if mode == "plain":
            pairs, last = ctx.sumcheck(d_f, n, ch)
            exp = co.sumcheck(f, ch)
            ok = (pairs == exp[:lg]).all() and (last == exp[lg, 1]).all()
        elif mode == "product":
            g = rand_fr(n, 30_000 + it)
            tr, lf, lg_ = ctx.sumcheck_product(d_f, ctx.to_device(g), n, ch)
            exp, elf, elg = co.sumcheck_product_rounds(f, g, ch)
            ok = (tr == exp).all() and (lf == elf).all() and (lg_ == elg).all()
        elif mode == "fold":
            npts = int(rng.integers(0, lg + 3))
            pts = rand_fr(max(npts, 1), 40_000 + it)[:npts]
            rounds = min(lg, npts)
            got = ctx.fold(d_f, n, pts).download((n >> rounds, 4))
            cur = f
            for i in range(rounds):
                cur = co.fold(cur, pts[i])
            ok = (got == cur).all()
            mode = f"fold[{npts}]"
        else:
            q, val = ctx.open_rounds(d_f, n, ch)
            eq, ev = co.open_quotients(f, ch)
            ok = (val == ev).all() and (n == 1 or (q.download((n - 1, 4)) == eq).all())
        if not ok:
            bad += 1
            print("MISMATCH", mode, "2^%d" % lg, "case", it, flush=True)
    if verbose:
        print(f"stress: {it} cases in {time.time() - t0:.0f} s, {bad} mismatches")
    return bad


if __name__ == "__main__":
    import zkhip, coracle

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    sys.exit(1 if run(zkhip.Ctx(0), coracle, seconds=budget, seed=int(os.environ.get("STRESS_SEED", "777"))) else 0)
