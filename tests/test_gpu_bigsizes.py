"""
Value-level parity at the sizes BASELINE config 4 actually runs (n = 24: product sumchecks on 2^24 .. 2^26 tables,
MSMs of 2^22 .. 2^25 points; hyperplonk/src/dhyperplonk.rs:198-553) -- bit-exact against the C oracle, which does a
2^24 product sumcheck in a few seconds on one host core and a 2^24-point MSM in well under a minute on the box's cores.

  * sumcheck_product / sumcheck / open_rounds / fold at 2^22 and 2^24 (and product + fold at 2^26 behind a memory guard)
    against coracle (dsumcheck.rs:6-90, dpoly_comm.rs:309-323, mle.rs:88-105);
  * the product sumcheck's DERIVED t1 (csrc/zk_fr.hip derive_t1: from 2^18 elements on t1 of every round after the first
    is p_{k-1}(r_{k-1}) - t0_k) against t1 = sum f_hi g_hi computed on the device in every round (test switch
    `sc_t1_device`), 2^18 .. 2^26;
  * MSM at 2^22 and 2^24 points, table-less and window-table path, against the chunked multi-thread oracle MSM
    (dmsm.rs:23);
  * a bounded run of the differential stress generator (tools/stress_sumcheck.py).
"""
import os
import sys

import numpy as np
import pytest

from helpers import jac_norm_to_affine, oracle_msm_chunked, rand_fr

pytestmark = pytest.mark.gpu


def _host_mem_gib():
    try:
        return os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
    except (ValueError, OSError):
        return 0


@pytest.mark.parametrize("lg", [22, 24])
def test_sumcheck_family_against_oracle(ctx, co, lg):
    N = 1 << lg
    f, g, ch = rand_fr(N, 100 + lg), rand_fr(N, 200 + lg), rand_fr(lg, 300 + lg)
    df, dg = ctx.to_device(f), ctx.to_device(g)
    tr, lf, lg_ = ctx.sumcheck_product(df, dg, N, ch)
    etr, elf, elg = co.sumcheck_product_rounds(f, g, ch)
    assert (tr == etr).all() and (lf == elf).all() and (lg_ == elg).all()
    pairs, last = ctx.sumcheck(df, N, ch)
    exp = co.sumcheck(f, ch)
    assert (pairs == exp[:lg]).all() and (last == exp[lg, 1]).all()
    q, val = ctx.open_rounds(df, N, ch)
    eq, ev = co.open_quotients(f, ch)
    assert (val == ev).all() and (q.download((N - 1, 4)) == eq).all()
    del q, eq
    # partial and full folds (fix_variable, mle.rs:88-105)
    cur = f
    for i in range(5):
        cur = co.fold(cur, ch[i])
    assert (ctx.fold(df, N, ch[:5]).download((N >> 5, 4)) == cur).all()
    assert (ctx.fold(df, N, ch).download((1, 4))[0] == ev).all()


def test_product_and_fold_2pow26_against_oracle(ctx, co):
    """the largest table of the n = 24 protocol (4M elements, c_sumcheck_product on V: dhyperplonk.rs:296-305)"""
    free, _ = ctx.mem_info()
    if free < (24 << 30) or _host_mem_gib() < 48:
        pytest.skip("needs ~10 GiB of HBM and ~20 GiB of host memory")
    lg = 26
    N = 1 << lg
    f, g, ch = rand_fr(N, 126), rand_fr(N, 226), rand_fr(lg, 326)
    df, dg = ctx.to_device(f), ctx.to_device(g)
    tr, lf, lg_ = ctx.sumcheck_product(df, dg, N, ch)
    etr, elf, elg = co.sumcheck_product_rounds(f, g, ch)
    assert (tr == etr).all() and (lf == elf).all() and (lg_ == elg).all()
    assert (ctx.fold(df, N, ch).download((1, 4))[0] == elf).all()
    assert (ctx.fold(dg, N, ch).download((1, 4))[0] == elg).all()


@pytest.mark.parametrize("lg", [10, 17, 18, 19, 20, 22, 24, 26])
def test_derived_t1_equals_device_t1(ctx, lg):
    """every t1 the host derives from the transcript == the device's own sum f_hi g_hi of that round"""
    N = 1 << lg
    if lg >= 26 and ctx.mem_info()[0] < (16 << 30):
        pytest.skip("needs ~8 GiB of HBM")
    f, g, ch = rand_fr(N, 400 + lg), rand_fr(N, 500 + lg), rand_fr(lg, 600 + lg)
    df, dg = ctx.to_device(f), ctx.to_device(g)
    del f, g
    a = ctx.sumcheck_product(df, dg, N, ch)
    ctx.dbg_tune("sc_t1_device", 1)
    try:
        b = ctx.sumcheck_product(df, dg, N, ch)
    finally:
        ctx.dbg_tune("sc_t1_device", 0)
    for x, y in zip(a, b):
        assert (x == y).all()
    c = ctx.sumcheck_product(df, dg, N, ch)  # the switch is really off again: same bits, derived path
    assert (c[0] == a[0]).all()


def test_dbg_tune_rejects_unknown_keys(ctx):
    import zkhip

    with pytest.raises(zkhip.ZkError):
        ctx.dbg_tune("no_such_knob", 1)


@pytest.mark.parametrize("lg", [22, 24])
def test_msm_against_chunked_oracle(ctx, co, lg):
    n = 1 << lg
    if lg >= 24 and (ctx.mem_info()[0] < (64 << 30) or (os.cpu_count() or 1) < 16):
        pytest.skip("the 2^24 comparison wants ~30 GiB of HBM for the window table and >= 16 host cores for the oracle")
    srs = ctx.srs_generate(0xC0FFEE + lg, 0x1234567 + lg, n)
    s = rand_fr(n, 700 + lg)
    # edge scalars inside a full-size run: 0, 1, r - 1 (Montgomery forms), a run of equal scalars (one long bucket)
    s[0] = 0
    s[1] = co.fr_to_mont(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
    rm1 = np.array([0xFFFFFFFF00000000, 0x53BDA402FFFE5BFE, 0x3339D80809A1D805, 0x73EDA753299D7D48], dtype=np.uint64)
    s[2] = co.fr_to_mont(rm1.reshape(1, 4))[0]
    s[1000:3000] = s[999]
    d = ctx.to_device(s)
    want = oracle_msm_chunked(srs.download(), s)
    r0 = ctx.msm_g1(srs, d, n)  # table-less path (GLV split, 17..19-bit windows)
    assert (jac_norm_to_affine(r0) == want).all()
    srs.precompute(0)  # window-table path (one bucket set)
    assert srs.table_window > 0
    r1 = ctx.msm_g1(srs, d, n)
    assert (r1 == r0).all()
    srs.free()


def test_bounded_differential_stress(ctx, co):
    """tools/stress_sumcheck.py's generator (random sizes, modes, partial folds) for a bounded number of cases"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_sumcheck

    assert stress_sumcheck.run(ctx, co, cases=600, seed=20260929) == 0


def _weighted_sums(canon: np.ndarray):
    """(sum_i s_i, sum_i i * s_i) as python ints for canonical scalars [n, 4] uint64 -- exact, vectorised: 32-bit half limbs,
    index blocks of 2^15 (t * half < 2^47, 2^15 terms per block sum)"""
    n = len(canon)
    halves = np.ascontiguousarray(canon).view(np.uint32).reshape(n, 8).astype(np.uint64)
    blk = 1 << 15
    pad = (-n) % blk
    if pad:
        halves = np.concatenate([halves, np.zeros((pad, 8), dtype=np.uint64)])
    h = halves.reshape(-1, blk, 8)
    t = np.arange(blk, dtype=np.uint64).reshape(1, blk, 1)
    a = h.sum(axis=1)        # [blocks, 8]  < 2^47
    b = (h * t).sum(axis=1)  # [blocks, 8]  < 2^62
    s0 = s1 = 0
    for c in range(h.shape[0]):
        av = sum(int(a[c, k]) << (32 * k) for k in range(8))
        bv = sum(int(b[c, k]) << (32 * k) for k in range(8))
        s0 += av
        s1 += c * blk * av + bv
    return s0, s1


@pytest.mark.parametrize("lg", [20, 24])
def test_msm_equals_its_closed_form(ctx, co, lg):
    """
    A check that shares NOTHING with the oracle's MSM: the synthetic SRS is P_i = (k0 + i k1) G (zk_srs_generate), so
        sum_i s_i P_i = (k0 sum_i s_i + k1 sum_i i s_i mod r) G
    -- two integer sums (numpy) and ONE scalar multiplication of the generator (python big-ints), against the library's
    MSM at full size, table-less and window-table path.
    """
    import pyoracle as po
    from helpers import pt_ints

    n = 1 << lg
    if lg >= 24 and ctx.mem_info()[0] < (64 << 30):
        pytest.skip("wants ~30 GiB of HBM for the window table")
    k0, k1 = 0xD15C0 + lg, 0x10C5 + 2 * lg
    srs = ctx.srs_generate(k0, k1, n)
    s = rand_fr(n, 4100 + lg)
    s0, s1 = _weighted_sums(co.fr_from_mont(s))
    want = po.g1_mul(po.G1_GEN, (k0 * s0 + k1 * s1) % po.R_MOD)
    d = ctx.to_device(s)
    assert pt_ints(jac_norm_to_affine(ctx.msm_g1(srs, d, n))) == want
    srs.precompute(0)
    assert pt_ints(jac_norm_to_affine(ctx.msm_g1(srs, d, n))) == want
    srs.free()


def _linear_table(ctx, coeffs):
    """device table of f(x) = c_0 + sum_k c_{k+1} x_k over {0,1}^n (index bit k <-> x_k), built by n doubling steps
    T[2^k .. 2^(k+1)) = T[0 .. 2^k) + c_{k+1} with zk_fr_axpb -- no sumcheck kernel involved"""
    from zkhip.field import fr_mont

    n = len(coeffs) - 1
    buf = ctx.alloc(32 << n)
    buf.upload(fr_mont(coeffs[0]).reshape(1, 4))
    zero = fr_mont(0)
    for k in range(n):
        ctx.fr_axpb(buf, buf, zero, fr_mont(coeffs[k + 1]), 1 << k, out=buf.at(32 << k))
    return buf


def _closed_form_product_sumcheck(a, b, ch):
    """transcript of sumcheck_product (dsumcheck.rs:28-90) for two LINEAR tables, by algebra only: over {0,1}^m,
    sum_x (c + R.x)(d + S.x) = 2^m c d + 2^(m-1) (c sum S + d sum R + R.S) + 2^(m-2) (sum R sum S - R.S)"""
    R_ = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    inv2 = pow(2, -1, R_)

    def S(c, R, d, Sg):
        m = len(R)
        dot = sum(x * y for x, y in zip(R, Sg))
        sr, ss = sum(R), sum(Sg)
        return pow(2, m, R_) * (c * d + (c * ss + d * sr + dot) * inv2 + (sr * ss - dot) * inv2 * inv2) % R_

    c, R, d, Sg = a[0], list(a[1:]), b[0], list(b[1:])
    rows = []
    for r in ch:
        ct, dt = R.pop(), Sg.pop()  # the top variable is folded first (lo = tab[..h], hi = tab[h..])
        rows.append((S(c, R, d, Sg), S(c + ct, R, d + dt, Sg), S(c + 2 * ct, R, d + 2 * dt, Sg)))
        c, d = (c + r * ct) % R_, (d + r * dt) % R_
    return rows, c, d


@pytest.mark.parametrize("lg", [12, 19, 22, 24, 26])
def test_sumcheck_family_equals_its_closed_form_on_linear_tables(ctx, lg):
    """
    Size-independent and oracle-independent: for f(x) = a_0 + sum a_k x_k and g alike (tables built on the device by doubling
    steps), every round tuple of sumcheck_product, the plain sumcheck's pairs, the fold and the open value have closed forms in
    the 2 (n + 1) coefficients and the challenges -- computed here with python big-ints -- at the table sizes the n = 24 proof runs.
    """
    import pyoracle as po
    from zkhip.field import fr_from_mont, fr_mont

    if lg >= 26 and ctx.mem_info()[0] < (16 << 30):
        pytest.skip("needs ~8 GiB of HBM")
    rng = po.SplitMix64(9000 + lg)
    a, b, ch = rng.fr_vec(lg + 1), rng.fr_vec(lg + 1), rng.fr_vec(lg)
    N = 1 << lg
    df, dg = _linear_table(ctx, a), _linear_table(ctx, b)
    chm = np.array([fr_mont(x) for x in ch], dtype=np.uint64)
    rows, cf, cg = _closed_form_product_sumcheck(a, b, ch)
    tr, lf, lg_ = ctx.sumcheck_product(df, dg, N, chm)
    got = [tuple(fr_from_mont(x) for x in t) for t in tr]
    assert got == rows
    assert (fr_from_mont(lf), fr_from_mont(lg_)) == (cf, cg)
    # plain sumcheck of f: (sum lo, sum hi) per round = the same closed form against the constant table g = 1
    one_rows, _, _ = _closed_form_product_sumcheck(a, [1] + [0] * lg, ch)
    pairs, last = ctx.sumcheck(df, N, chm)
    assert [tuple(fr_from_mont(x) for x in p) for p in pairs] == [(t0, t1) for t0, t1, _ in one_rows]
    assert fr_from_mont(last) == cf
    # fix_variable with every point, and the value of open (dpoly_comm.rs:309-325): f at the challenge point
    assert fr_from_mont(ctx.fold(df, N, chm).download((1, 4))[0]) == cf
    if lg <= 24:
        q, val = ctx.open_rounds(dg, N, chm)
        assert fr_from_mont(val) == cg
        # q_0 = hi - lo of a linear table is the constant b_n (the top variable's coefficient)
        assert [fr_from_mont(x) for x in q.download((2, 4))] == [b[lg] % po.R_MOD] * 2


@pytest.mark.parametrize("lg", [10, 23])
def test_product_tree_closed_form_on_a_constant_table(ctx, lg):
    """acc_product (dacc_product.rs:30-57) of x = [c; N]: level l of the tree is c^(2^l) throughout, the last entry the forced 0 --
    at N = M / 2 of the n = 24 proof, with no oracle in the loop"""
    from zkhip.field import R_MOD, fr_from_mont, fr_mont

    N, c = 1 << lg, 0x1234567890ABCDEF0FEDCBA987654321
    x = ctx.fr_axpb(None, ctx.alloc(32 * N), fr_mont(0), fr_mont(c), N)  # 0 * garbage + c
    tree = ctx.product_tree(x, N)
    off, level = 0, 0
    while off < 2 * N - 1:
        cnt = N >> level
        want = pow(c, 1 << level, R_MOD)
        for pos in {off, off + cnt // 2, off + cnt - 1} - {2 * N - 1}:
            assert fr_from_mont(tree.download((1, 4), offset=32 * pos)[0]) == want, (level, pos)
        off += cnt
        level += 1
    assert fr_from_mont(tree.download((1, 4), offset=32 * (2 * N - 1))[0]) == 0
