"""
G2 MSM on the GPU (zk_msm_g2 / zk_msm_g2_batch: the MSM pipeline instantiated over Fq2) against the oracle's
affine G2 arithmetic.  `d_msm` / `G::msm` are generic over CurveGroup (dmsm.rs:9,23); the reference's G2 points
live in powers_of_g2 (dpoly_comm.rs:27,59-62).  Large inputs use bases with KNOWN discrete logs
(P_i = (k0 + i k1) G2), so the expected result is one scalar multiplication: (sum_i s_i (k0 + i k1)) G2.
"""
import numpy as np
import pytest

import pyoracle as po
from helpers import rand_fr

pytestmark = pytest.mark.gpu

ONE_M = np.array(po.fq_to_mont_limbs(1), dtype=np.uint64)


def _aff(j36):
    """normalised Jacobian [36] -> oracle point"""
    j = np.asarray(j36, dtype=np.uint64).reshape(36)
    if not j[24:].any():
        return None
    assert (j[24:30] == ONE_M).all() and not j[30:].any(), "library results must be normalised (z = 1)"
    return po.g2_from_mont_limbs(j[:24])


def _seq(n, k0, k1):
    """P_i = (k0 + i k1) G2 as [n, 24] Montgomery limbs, by repeated affine addition"""
    step = po.g2_mul(po.G2_GEN, k1)
    cur = po.g2_mul(po.G2_GEN, k0)
    out = []
    for _ in range(n):
        out.append(po.g2_to_mont_limbs(cur))
        cur = po.g2_add(cur, step)
    return np.array(out, dtype=np.uint64)


def _ints(a):
    return [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]


def test_g2_point_formulas(ctx):
    """device XYZZ formulas over Fq2 on pairs of points, including the doubling / cancellation branches"""
    rng = po.SplitMix64(31)
    n = 24
    P = [po.g2_mul(po.G2_GEN, rng.fr()) for _ in range(n)]
    Q = [po.g2_mul(po.G2_GEN, rng.fr()) for _ in range(n)]
    Q[3], Q[4] = P[3], po.g2_neg(P[4])  # p + p (mixed-addition doubling), p - p
    dp = ctx.to_device(np.array([po.g2_to_mont_limbs(X) for X in P], dtype=np.uint64))
    dq = ctx.to_device(np.array([po.g2_to_mont_limbs(X) for X in Q], dtype=np.uint64))
    add, neg = po.g2_add, po.g2_neg
    exp = {0: lambda p, q: add(p, q), 1: lambda p, q: add(add(p, q), p), 2: lambda p, q: add(add(p, q), add(p, q)), 3: lambda p, q: add(p, neg(q)),
           4: lambda p, q: add(add(p, q), add(p, q)), 5: lambda p, q: None}
    for mode, f in exp.items():
        got = ctx.dbg_g2_op(mode, dp, dq, n)
        for i in range(n):
            assert _aff(got[i]) == f(P[i], Q[i]), (mode, i)


@pytest.mark.parametrize("n", [1, 2, 7, 64, 300])
def test_msm_g2_small_against_oracle_msm(ctx, n):
    rng = po.SplitMix64(3000 + n)
    pts = [po.g2_mul(po.G2_GEN, rng.fr()) for _ in range(n)]
    sc = rand_fr(n, 40 + n)
    srs = ctx.srs_register_g2(np.array([po.g2_to_mont_limbs(P) for P in pts], dtype=np.uint64))
    assert len(srs) == n and (srs.download() == np.array([po.g2_to_mont_limbs(P) for P in pts], dtype=np.uint64)).all()
    got = _aff(ctx.msm_g2(srs, ctx.to_device(sc), n))
    assert got == po.g2_msm(pts, _ints(sc))


@pytest.mark.parametrize("lg", [10, 13, 15])
def test_msm_g2_known_discrete_logs(ctx, lg):
    n = 1 << lg
    k0, k1 = 0x1234567 + lg, 0xABCDEF01
    bases = _seq(n, k0, k1)
    sc = rand_fr(n, 900 + lg)
    srs = ctx.srs_register_g2(bases)
    got = _aff(ctx.msm_g2(srs, ctx.to_device(sc), n))
    e = sum(s * (k0 + i * k1) for i, s in enumerate(_ints(sc))) % po.R_MOD
    assert got == po.g2_mul(po.G2_GEN, e)
    # a sub-range with an offset
    got = _aff(ctx.msm_g2(srs, ctx.to_device(sc[: n // 2]), n // 2, offset=5))
    e = sum(s * (k0 + (i + 5) * k1) for i, s in enumerate(_ints(sc[: n // 2]))) % po.R_MOD
    assert got == po.g2_mul(po.G2_GEN, e)


def test_msm_g2_edge_cases(ctx):
    """zero scalars, r - 1, infinity among the bases, repeated bases (doubling path), cancellation, empty input"""
    rng = po.SplitMix64(5)
    P, Q = po.g2_mul(po.G2_GEN, rng.fr()), po.g2_mul(po.G2_GEN, rng.fr())
    pts = [P, P, Q, None, po.g2_neg(P), Q, P, P]
    sc = [5, 5, 0, 77, 5, po.R_MOD - 1, 1, 2]
    to_m = lambda xs: np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)
    srs = ctx.srs_register_g2(np.array([po.g2_to_mont_limbs(X) for X in pts], dtype=np.uint64))
    assert _aff(ctx.msm_g2(srs, ctx.to_device(to_m(sc)), 8)) == po.g2_msm(pts, sc)
    assert _aff(ctx.msm_g2(srs, ctx.to_device(to_m([1, po.R_MOD - 1])), 2)) is None  # P - P
    assert _aff(ctx.msm_g2(srs, ctx.to_device(to_m([0] * 8)), 8)) is None
    assert _aff(ctx.msm_g2(srs, ctx.to_device(to_m([1])), 0)) is None
    # 193-byte stride with the infinity flag of the Rust struct (padded to 200)
    rec = np.zeros((3, 200), dtype=np.uint8)
    for i, X in enumerate([P, Q, P]):
        rec[i, :192] = np.array(po.g2_to_mont_limbs(X), dtype=np.uint64).view(np.uint8)
    rec[1, 192] = 1  # Q flagged as infinity
    s2 = ctx.srs_register_g2(rec, stride=200)
    assert _aff(ctx.msm_g2(s2, ctx.to_device(to_m([3, 9, 4])), 3)) == po.g2_mul(P, 7)


def test_msm_g2_batch_and_group_mismatch(ctx):
    import zkhip
    from helpers import synthetic_bases

    k0, k1 = 99, 12345
    bases = _seq(1 << 11, k0, k1)
    srs = ctx.srs_register_g2(bases)
    lens = [1 << 11, 700, 3]
    scs = [rand_fr(m, 70 + m) for m in lens]
    outs = ctx.msm_g2_batch([srs] * 3, [ctx.to_device(s) for s in scs], lens)
    for m, s, o in zip(lens, scs, outs):
        e = sum(v * (k0 + i * k1) for i, v in enumerate(_ints(s))) % po.R_MOD
        assert _aff(o) == po.g2_mul(po.G2_GEN, e)
    g1, _ = synthetic_bases(16, 1)
    with pytest.raises(zkhip.ZkError):  # a G1 vector handed to the G2 entry point
        ctx.msm_g2(ctx.srs_register(g1), ctx.to_device(scs[2]), 3)
    with pytest.raises(zkhip.ZkError):
        ctx.msm_g1(srs, ctx.to_device(scs[2]), 3)


@pytest.mark.parametrize("lg,c", [(6, 0), (11, 0), (14, 0), (14, 9), (15, 16)])
def test_msm_g2_window_table_matches_table_less_path(ctx, lg, c):
    """zk_srs_precompute on a G2 level (table of 2^{o_w} P_i over Fq2): same result bits as the table-less path, which the
    tests above pin to the oracle; offsets and batches included"""
    n = 1 << lg
    k0, k1 = 0x7654321 + lg, 0x1F2E3D
    srs = ctx.srs_register_g2(_seq(n, k0, k1))
    sc = rand_fr(n, 1200 + lg)
    sc[0] = 0
    sc[1] = sc[2]
    d = ctx.to_device(sc)
    want = ctx.msm_g2(srs, d, n)
    want_off = ctx.msm_g2(srs, ctx.to_device(sc[: n // 2]), n // 2, offset=3)
    srs.precompute(c)
    assert srs.table_window == (c or srs.table_window) and srs.table_window > 0
    assert (ctx.msm_g2(srs, d, n) == want).all()
    assert (ctx.msm_g2(srs, ctx.to_device(sc[: n // 2]), n // 2, offset=3) == want_off).all()
    e = sum(s * (k0 + i * k1) for i, s in enumerate(_ints(sc))) % po.R_MOD
    assert _aff(want) == po.g2_mul(po.G2_GEN, e)
    outs = ctx.msm_g2_batch([srs, srs], [d, d], [n, n // 4])
    assert (outs[0] == want).all()
    e4 = sum(s * (k0 + i * k1) for i, s in enumerate(_ints(sc[: n // 4]))) % po.R_MOD
    assert _aff(outs[1]) == po.g2_mul(po.G2_GEN, e4)
