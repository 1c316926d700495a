"""Error behaviour of the C ABI: the reference's assert!s / Err(..) become negative status codes, nothing aborts."""
import ctypes

import numpy as np
import pytest

import zkhip
from helpers import rand_fr, synthetic_bases
from zkhip._lib import ZK_ERR_INVALID, ZK_ERR_LENGTH

pytestmark = pytest.mark.gpu


def test_non_power_of_two_and_null_arguments(ctx):
    lib, h = ctx.lib, ctx.h
    d = ctx.to_device(rand_fr(12, 1))
    ch = rand_fr(4, 2)
    out = np.zeros((4, 3, 4), dtype=np.uint64)
    last = np.zeros(4, dtype=np.uint64)
    # table lengths must be powers of two (dpoly_comm.rs:239-240, dsumcheck.rs `trailing_zeros`)
    assert lib.zk_sumcheck(h, d.ptr, 12, ch.ctypes.data, out.ctypes.data, last.ctypes.data) == ZK_ERR_INVALID
    assert b"power of two" in lib.zk_last_error(h)
    assert lib.zk_sumcheck_product(h, d.ptr, d.ptr, 0, ch.ctypes.data, out.ctypes.data, last.ctypes.data, last.ctypes.data) == ZK_ERR_INVALID
    assert lib.zk_product_tree(h, d.ptr, 6, d.ptr) == ZK_ERR_INVALID
    assert lib.zk_fold(h, d.ptr, 10, ch.ctypes.data, 2, d.ptr) == ZK_ERR_INVALID
    # null pointers
    assert lib.zk_sumcheck(h, None, 8, ch.ctypes.data, out.ctypes.data, last.ctypes.data) == ZK_ERR_INVALID
    assert lib.zk_msm_g1(h, None, 0, d.ptr, 4, out.ctypes.data) == ZK_ERR_INVALID
    assert lib.zk_fr_batch_div(h, d.ptr, d.ptr, d.ptr, 4) == ZK_ERR_INVALID  # out must not alias
    # the context stays usable after errors
    pairs, _ = ctx.sumcheck(d, 8, ch[:3])
    assert pairs.shape == (3, 2, 4)


def test_msm_more_scalars_than_bases(ctx):
    bases, _ = synthetic_bases(16, 3)
    srs = ctx.srs_register(bases)
    d = ctx.to_device(rand_fr(32, 4))
    with pytest.raises(zkhip.MsmLengthError):
        ctx.msm_g1(srs, d, 32)
    with pytest.raises(zkhip.MsmLengthError):
        ctx.msm_g1(srs, d, 8, offset=12)
    out = np.zeros(18, dtype=np.uint64)
    assert ctx.lib.zk_msm_g1(ctx.h, srs.h, 0, d.ptr, 17, out.ctypes.data) == ZK_ERR_LENGTH
    # ... and a valid call still works afterwards
    assert ctx.msm_g1(srs, d, 16).shape == (18,)


def test_fold_with_more_points_than_variables(ctx, co):
    """fix_variable folds min(n, points) times (mle.rs:94)"""
    t = rand_fr(8, 5)
    pts = rand_fr(6, 6)
    got = ctx.fold(ctx.to_device(t), 8, pts).download((1, 4))
    cur = t
    for i in range(3):
        cur = co.fold(cur, pts[i])
    assert (got == cur).all()


def test_single_element_tables(ctx):
    """len = 1: zero rounds; the last element is the table itself"""
    t = rand_fr(1, 7)
    pairs, last = ctx.sumcheck(ctx.to_device(t), 1, np.zeros((0, 4), dtype=np.uint64))
    assert pairs.shape[0] == 0 and (last == t[0]).all()
    tr, lf, lg = ctx.sumcheck_product(ctx.to_device(t), ctx.to_device(t), 1, np.zeros((0, 4), dtype=np.uint64))
    assert tr.shape[0] == 0 and (lf == t[0]).all() and (lg == t[0]).all()
    q, v = ctx.open_rounds(ctx.to_device(t), 1, np.zeros((0, 4), dtype=np.uint64))
    assert (v == t[0]).all()


def test_ctx_on_every_visible_gpu_in_one_process(co):
    """
    one process, one ctx per GPU (the reference's task-per-party model): every kernel attribute the library sets
    (dynamic LDS of the sumcheck stages, of the staged scatter) must hold on EVERY device, not only the first one
    that launched.  Runs the LDS-heavy paths on each visible GPU (one on a 1-GPU box).
    """
    import zkhip
    from helpers import rand_fr

    ndev = zkhip.lib().zk_device_count()
    assert ndev >= 1
    f, g, ch = rand_fr(1 << 12, 1), rand_fr(1 << 12, 2), rand_fr(12, 3)
    etr, elf, elg = co.sumcheck_product_rounds(f, g, ch)
    ctxs = [zkhip.Ctx(d) for d in range(ndev)]
    try:
        for c in ctxs:
            tr, lf, lg = c.sumcheck_product(c.to_device(f), c.to_device(g), 1 << 12, ch)
            assert (tr == etr).all() and (lf == elf).all() and (lg == elg).all()
            pairs, last = c.sumcheck(c.to_device(f), 1 << 12, ch)
            assert (pairs == co.sumcheck(f, ch)[:12]).all()
    finally:
        for c in ctxs:
            c.close()
