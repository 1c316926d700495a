"""
Parity at BASELINE.json's full sizes (2^20 shares): direct bit-exact comparison with the C oracle
where it finishes in seconds, plus size-independent properties (verifier accepts, linearity,
round trips) for everything else.
"""
import numpy as np
import pytest

from helpers import jac_norm_to_affine, rand_fr

pytestmark = pytest.mark.gpu
N20 = 1 << 20


def test_sumcheck_product_2pow20_bit_exact(ctx, co):
    f, g, ch = rand_fr(N20, 11), rand_fr(N20, 12), rand_fr(20, 13)
    tr, lf, lg = ctx.sumcheck_product(ctx.to_device(f), ctx.to_device(g), N20, ch)
    etr, elf, elg = co.sumcheck_product_rounds(f, g, ch)
    assert (tr == etr).all() and (lf == elf).all() and (lg == elg).all()


def test_sumcheck_2pow20_bit_exact(ctx, co):
    f, ch = rand_fr(N20, 21), rand_fr(20, 22)
    pairs, last = ctx.sumcheck(ctx.to_device(f), N20, ch)
    exp = co.sumcheck(f, ch)
    assert (pairs == exp[:20]).all() and (last == exp[20, 1]).all()


def test_open_fold_tree_2pow20(ctx, co):
    f, pt = rand_fr(N20, 31), rand_fr(20, 32)
    q, val = ctx.open_rounds(ctx.to_device(f), N20, pt)
    eq, ev = co.open_quotients(f, pt)
    assert (val == ev).all() and (q.download((N20 - 1, 4)) == eq).all()
    # fold to a value == open value (fix_variable with all n points, mle.rs:88-105)
    assert (ctx.fold(ctx.to_device(f), N20, pt).download((1, 4))[0] == ev).all()
    x = rand_fr(1 << 19, 33)
    assert (ctx.product_tree(ctx.to_device(x), 1 << 19).download((N20, 4)) == co.product_tree(x)).all()


def test_msm_2pow20_vs_oracle_and_properties(ctx, co):
    """one full-size MSM checked against the single-threaded C oracle (~15-20 s of CPU), then properties"""
    srs = ctx.srs_generate(0xABCDEF, 0x123457, N20)
    s1 = rand_fr(N20, 41)
    d1 = ctx.to_device(s1)
    r1 = ctx.msm_g1(srs, d1, N20)
    assert (jac_norm_to_affine(r1) == co.msm_g1(srs.download(), s1)).all()
    # the precomputed-table path and a different window size agree bit for bit (normalised output)
    ctx.msm_set_window(14)
    try:
        assert (ctx.msm_g1(srs, d1, N20) == r1).all()
    finally:
        ctx.msm_set_window(0)
    srs.precompute(18)
    assert (ctx.msm_g1(srs, d1, N20) == r1).all()
    # linearity: MSM(s1) + MSM(s2) == MSM(s1 + s2); MSM(alpha * s1) == alpha * MSM(s1)
    s2 = rand_fr(N20, 42)
    d2 = ctx.to_device(s2)
    r2 = ctx.msm_g1(srs, d2, N20)
    r3 = ctx.msm_g1(srs, ctx.fr_add(d1, d2, N20), N20)
    assert (co.g1_add_affine(jac_norm_to_affine(r1), jac_norm_to_affine(r2)) == jac_norm_to_affine(r3)).all()
    alpha = rand_fr(1, 43)[0]
    zero = np.zeros(4, dtype=np.uint64)
    scaled = ctx.fr_axpb(ctx.to_device(np.zeros((N20, 4), dtype=np.uint64)), d1, alpha, zero, N20)
    r4 = ctx.msm_g1(srs, scaled, N20)
    assert (jac_norm_to_affine(r4) == co.g1_mul_affine(jac_norm_to_affine(r1), co.fr_from_mont(alpha.reshape(1, 4))[0])).all()
