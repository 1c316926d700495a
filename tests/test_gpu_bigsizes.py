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
