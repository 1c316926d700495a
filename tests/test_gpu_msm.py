"""GPU parity: G1 MSM (zk_msm_g1 / zk_msm_g1_host) vs the CPU oracle, bit-exact on the affine point."""
import numpy as np
import pytest

from helpers import jac_norm_to_affine, pt_mont, rand_fr, synthetic_bases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 257, 1000, 4096, 1 << 14])
def test_msm_matches_oracle(ctx, co, n):
    bases, _ = synthetic_bases(n, 40 + n)
    scalars = rand_fr(n, 50 + n)
    srs = ctx.srs_register(bases)
    got = ctx.msm_g1(srs, ctx.to_device(scalars), n)
    assert (jac_norm_to_affine(got) == co.msm_g1(bases, scalars)).all()


@pytest.mark.parametrize("c", [2, 3, 4, 5, 7, 8, 11, 13, 14, 15, 16, 17, 18, 19, 20])
def test_msm_all_window_sizes(ctx, co, c):
    n = 3000
    bases, _ = synthetic_bases(n, 77)
    scalars = rand_fr(n, 78)
    srs = ctx.srs_register(bases)
    exp = co.msm_g1(bases, scalars)
    ctx.msm_set_window(c)
    try:
        got = ctx.msm_g1(srs, ctx.to_device(scalars), n)
    finally:
        ctx.msm_set_window(0)
    assert (jac_norm_to_affine(got) == exp).all()


def test_msm_edge_cases(ctx, co):
    import pyoracle as po

    n = 64
    bases, _ = synthetic_bases(n, 91)
    scalars = rand_fr(n, 92)
    # zero scalars, scalar = 1, scalar = r-1, repeated points, opposite points, infinity bases
    scalars[0] = 0
    scalars[1] = po.fr_to_mont_limbs(1)
    scalars[2] = po.fr_to_mont_limbs(po.R_MOD - 1)
    bases[4] = bases[3]
    scalars[4] = scalars[3]  # same point, same scalar -> same bucket (doubling path)
    bases[6] = bases[5]
    scalars[6] = co.fr_sub(np.zeros((1, 4), dtype=np.uint64), scalars[5:6])[0]  # s and -s on the same point
    bases[7] = 0  # infinity
    srs = ctx.srs_register(bases)
    got = ctx.msm_g1(srs, ctx.to_device(scalars), n)
    assert (jac_norm_to_affine(got) == co.msm_g1(bases, scalars)).all()
    # all-zero scalars -> infinity; n = 0 -> infinity
    z = np.zeros((n, 4), dtype=np.uint64)
    assert not jac_norm_to_affine(ctx.msm_g1(srs, ctx.to_device(z), n)).any()
    assert not jac_norm_to_affine(ctx.msm_g1(srs, ctx.to_device(z), 0)).any()
    # all scalars equal (one bucket per window holds everything)
    same = np.tile(scalars[9], (n, 1))
    assert (jac_norm_to_affine(ctx.msm_g1(srs, ctx.to_device(same), n)) == co.msm_g1(bases, same)).all()


def test_msm_endomorphism_split_boundaries(ctx, co):
    """
    The library splits every scalar as k = k1 + k2*lambda (lambda = z^2 - 1) and runs on P_i, phi(P_i):
    scalars sitting on the boundaries of that split (multiples of lambda, quotient / remainder extremes,
    carries of the signed recoding at 2^127 / 2^128) must still give the oracle's point.
    """
    import pyoracle as po

    lam = 0xAC45A4010001A40200000000FFFFFFFF
    r = po.R_MOD
    assert (lam * lam + lam + 1) % r == 0
    special = [0, 1, 2, lam - 1, lam, lam + 1, 2 * lam - 1, 2 * lam, lam * lam % r, (lam * lam - 1) % r, r - 1, r - 2, r - lam,
               r - lam - 1, (r // lam) * lam, (r // lam) * lam - 1, (1 << 127) - 1, 1 << 127, (1 << 127) + 1, (1 << 128) - 1, 1 << 128,
               (1 << 128) + 1, (lam << 127) % r, (1 << 254), r >> 1, (r >> 1) + 1, lam * ((1 << 127) - 1) % r,
               ((1 << 128) - 1) * lam % r, (lam - 1) + (lam) * lam if (lam - 1) + lam * lam < r else lam - 1]
    n = len(special)
    bases, _ = synthetic_bases(n, 191)
    scalars = np.array([po.fr_to_mont_limbs(v % r) for v in special], dtype=np.uint64)
    srs = ctx.srs_register(bases)
    d = ctx.to_device(scalars)
    exp = co.msm_g1(bases, scalars)
    for c in (0, 5, 9, 13, 16, 17):
        ctx.msm_set_window(c)
        try:
            got = ctx.msm_g1(srs, d, n)
        finally:
            ctx.msm_set_window(0)
        assert (jac_norm_to_affine(got) == exp).all(), c
    # one scalar at a time: the point itself is checked, not just the sum
    for i in range(n):
        got = ctx.msm_g1(srs, ctx.to_device(scalars[i : i + 1]), 1, offset=i)
        assert (jac_norm_to_affine(got) == co.msm_g1(bases[i : i + 1], scalars[i : i + 1])).all(), hex(special[i])


def test_msm_host_dropin_and_length_error(ctx, co):
    import zkhip

    n = 200
    bases, _ = synthetic_bases(n, 93)
    scalars = rand_fr(n, 94)
    exp = co.msm_g1(bases, scalars)
    assert (jac_norm_to_affine(ctx.msm_g1_host(bases, scalars)) == exp).all()
    # Rust struct layout: stride 104 with the infinity flag byte
    b104 = np.zeros((n, 104), dtype=np.uint8)
    b104[:, :96] = bases.view(np.uint8).reshape(n, 96)
    b104[10, 96] = 1  # flagged infinity (coordinates left in place, as ark does not clear them)
    bases_inf = bases.copy()
    bases_inf[10] = 0
    assert (jac_norm_to_affine(ctx.msm_g1_host(b104, scalars, stride=104)) == co.msm_g1(bases_inf, scalars)).all()
    # G::msm -> Err(min_len) when the slices differ in length (ark-ec 0.4.2), dmsm.rs:23
    with pytest.raises(zkhip.MsmLengthError) as ei:
        ctx.msm_g1_host(bases[:150], scalars)
    assert ei.value.min_len == 150


def test_srs_generate_matches_oracle_points(ctx, co):
    import pyoracle as po

    n = 1500
    bases, (k0, k1) = synthetic_bases(n, 95)
    srs = ctx.srs_generate(k0, k1, n)
    assert (srs.download() == bases).all()


def test_msm_linearity_property_large(ctx, co):
    """size-independent property at 2^18: MSM(b, s1) + MSM(b, s2) == MSM(b, s1 + s2)"""
    n = 1 << 18
    srs = ctx.srs_generate(12345, 67891, n)
    s1, s2 = rand_fr(n, 96), rand_fr(n, 97)
    d1, d2 = ctx.to_device(s1), ctx.to_device(s2)
    d3 = ctx.fr_add(d1, d2, n)
    r1 = jac_norm_to_affine(ctx.msm_g1(srs, d1, n))
    r2 = jac_norm_to_affine(ctx.msm_g1(srs, d2, n))
    r3 = jac_norm_to_affine(ctx.msm_g1(srs, d3, n))
    assert (co.g1_add_affine(r1, r2) == r3).all()


def test_msm_skewed_scalars_long_buckets(ctx, co):
    """skewed digit distributions (few distinct scalars -> giant buckets) take the cooperative fix-up path"""
    n = 1 << 14
    bases, _ = synthetic_bases(n, 120)
    few = rand_fr(3, 121)
    scalars = few[np.arange(n) % 3].copy()  # only three distinct scalars: ~n/3 points per bucket per window
    scalars[::5] = rand_fr(n, 122)[::5]
    srs = ctx.srs_register(bases)
    exp = co.msm_g1(bases, scalars)
    for c in (0, 9, 12):
        ctx.msm_set_window(c)
        try:
            got = ctx.msm_g1(srs, ctx.to_device(scalars), n)
        finally:
            ctx.msm_set_window(0)
        assert (jac_norm_to_affine(got) == exp).all(), c
    # small scalars only (high windows all zero, low windows dense)
    small = np.zeros((n, 4), dtype=np.uint64)
    small[:, 0] = np.arange(n) % 7
    sm = co.fr_to_mont(small)
    assert (jac_norm_to_affine(ctx.msm_g1(srs, ctx.to_device(sm), n)) == co.msm_g1(bases, sm)).all()


def test_msm_batch_matches_individual(ctx, co):
    """zk_msm_g1_batch: heterogeneous sizes (the halving batch of c_open / open) in one pass"""
    sizes = [1 << 13, 1 << 12, 1 << 11, 700, 64, 33, 8, 4, 2, 1, 0, 1 << 13]
    srs, scal, exp = [], [], []
    for k, n in enumerate(sizes):
        bases, _ = synthetic_bases(max(n, 1), 600 + k)
        sc = rand_fr(max(n, 1), 700 + k)
        srs.append(ctx.srs_register(bases))
        scal.append(ctx.to_device(sc))
        exp.append(co.msm_g1(bases[:n], sc[:n]) if n else np.zeros(12, dtype=np.uint64))
    got = ctx.msm_g1_batch(srs, scal, sizes)
    for k in range(len(sizes)):
        assert (jac_norm_to_affine(got[k]) == exp[k]).all(), (k, sizes[k])
    # offsets into one SRS: sub-ranges of the same level
    bases, _ = synthetic_bases(4096, 650)
    big = ctx.srs_register(bases)
    sc = rand_fr(4096, 651)
    d = ctx.to_device(sc)
    got = ctx.msm_g1_batch([big, big], [d, d.at(32 * 1024)], [1024, 512], offsets=[0, 1024])
    assert (jac_norm_to_affine(got[0]) == co.msm_g1(bases[:1024], sc[:1024])).all()
    assert (jac_norm_to_affine(got[1]) == co.msm_g1(bases[1024:1536], sc[1024:1536])).all()


def test_window_table_record_layouts_give_the_same_points(ctx):
    """zk_srs_precompute_layout: packed 96-byte records and one record per 128-byte line -- same MSM
    results with and without the table, also for a sub-range of the level and in a batch that mixes both layouts; bad values are refused"""
    import zkhip

    n = 1 << 13
    srs_a, srs_b = ctx.srs_generate(321, 654, n), ctx.srs_generate(321, 654, n)
    sc = ctx.to_device(rand_fr(n, 77))
    ref = ctx.msm_g1(srs_a, sc, n)
    ref_sub = ctx.msm_g1(srs_a, sc, 1000, offset=123)
    srs_a.precompute(0, record_bytes=96)
    srs_b.precompute(0, record_bytes=128)
    assert srs_a.table_record == 96 and srs_b.table_record == 128 and srs_a.table_window == srs_b.table_window > 0
    for srs in (srs_a, srs_b):
        assert (ctx.msm_g1(srs, sc, n) == ref).all()
        assert (ctx.msm_g1(srs, sc, 1000, offset=123) == ref_sub).all()
    got = ctx.msm_g1_batch([srs_a, srs_b, srs_b], [sc, sc, sc], [n, n, 1000])
    assert (got[0] == ref).all() and (got[1] == ref).all() and (got[2] == ctx.msm_g1(srs_a, sc, 1000)).all()
    with pytest.raises(zkhip.ZkError):
        srs_a.precompute(0, record_bytes=100)
    srs_b.precompute(0)  # (rebuilt in the default layout: by free memory -- 128-byte records while the table leaves >= 60 % of the device free)
    free, total = ctx.mem_info()
    assert srs_b.table_record == (128 if free >= 0.65 * total else srs_b.table_record) and srs_b.table_record in (96, 128)
    assert (ctx.msm_g1(srs_b, sc, n) == ref).all()
    # a table wider than 20 bits (the automatic pick stops there; 22 = the sort's 1 024 partitions x 2 048 buckets) -- same points
    srs_b.precompute(22)
    assert srs_b.table_window == 22 and (ctx.msm_g1(srs_b, sc, n) == ref).all()
    with pytest.raises(zkhip.ZkError):
        srs_b.precompute(23)


def test_msm_batch_of_window_table_items_of_many_sizes(ctx, co):
    """
    the short items of a proof's passes (dpoly_comm.rs:401-464: an open commits quotients of 2^k, 2^(k-1), .. points): levels of
    64 .. 2^15 points with the library's own table widths (two common widths up to 2^14 points), items shorter than their level and
    at offsets, all in ONE batch -- items of one width and up to 2^14 points share a class whose rows are as long as its longest
    item.  Same bits as the oracle, with the merged classes (default) and with a class per size / a width per size.
    """
    sizes = [64, 100, 128, 500, 1 << 10, 1500, 1 << 11, 3000, 1 << 12, 1 << 13, 10000, 1 << 14, 20000, 1 << 15, 1 << 12, 64, 1 << 14]
    use = [64, 70, 128, 257, 1 << 10, 1, 1 << 11, 2999, 4000, 1 << 13, 9999, 1 << 14, 16385, 1 << 15, 1 << 12, 3, 12345]
    off = [0, 30, 0, 100, 0, 1499, 0, 1, 96, 0, 1, 0, 3615, 0, 0, 61, 4039]
    for knobs in ((), (("msm_size_class_min", 0), ("msm_small_table_widths", 0)), (("msm_size_class_min", 11),), (("msm_size_class_min", 30),)):
        for k, v in knobs:
            ctx.dbg_tune(k, v)
        try:
            srs, scal, exp = [], [], []
            for i, n in enumerate(sizes):
                bases, _ = synthetic_bases(n, 900 + i)
                sc = rand_fr(n, 950 + i)
                lv = ctx.srs_register(bases).precompute(0)
                srs.append(lv)
                scal.append(ctx.to_device(sc).at(32 * off[i]))
                exp.append(co.msm_g1(bases[off[i] : off[i] + use[i]], sc[off[i] : off[i] + use[i]]))
            widths = {len(lv): lv.table_window for lv in srs}
            if not knobs:
                assert widths[64] == widths[1 << 10] == 12 and widths[1 << 11] == widths[1 << 14] == 14 and widths[1 << 15] == 17, widths
            got = ctx.msm_g1_batch(srs, scal, use, offsets=off)
            for i in range(len(sizes)):
                assert (jac_norm_to_affine(got[i]) == exp[i]).all(), (knobs, i, sizes[i], use[i], off[i])
            for lv in srs:
                lv.free()
        finally:
            ctx.dbg_tune("msm_size_class_min", 16)
            ctx.dbg_tune("msm_small_table_widths", 1)


@pytest.mark.parametrize("n,c", [(1, 0), (100, 5), (5000, 0), (5000, 13), (1 << 15, 0), (1 << 15, 16)])
def test_msm_precomputed_srs_matches_oracle(ctx, co, n, c):
    """shared-bucket mode (zk_srs_precompute): same result as the oracle, also on sub-ranges and in batches"""
    bases, _ = synthetic_bases(n, 800 + n)
    bases[n // 2] = 0  # an infinity base must stay infinity in every table copy
    scalars = rand_fr(n, 801 + n)
    srs = ctx.srs_register(bases).precompute(c)
    d = ctx.to_device(scalars)
    assert (jac_norm_to_affine(ctx.msm_g1(srs, d, n)) == co.msm_g1(bases, scalars)).all()
    if n >= 100:
        off, m = n // 4, n // 2
        got = ctx.msm_g1(srs, d.at(32 * off), m, offset=off)
        assert (jac_norm_to_affine(got) == co.msm_g1(bases[off : off + m], scalars[off : off + m])).all()
        plain = ctx.srs_register(bases)
        got = ctx.msm_g1_batch([srs, plain, srs], [d, d, d.at(32 * off)], [n, n, m], offsets=[0, 0, off])
        exp = co.msm_g1(bases, scalars)
        assert (jac_norm_to_affine(got[0]) == exp).all() and (jac_norm_to_affine(got[1]) == exp).all()
        assert (jac_norm_to_affine(got[2]) == co.msm_g1(bases[off : off + m], scalars[off : off + m])).all()


def test_msm_2pow22_window19_linearity(ctx, co):
    """
    above 2^22 the 19-bit layout (7 windows) takes over; too large for the CPU oracle in seconds, so:
    MSM(b, s1 + s2) == MSM(b, s1) + MSM(b, s2), and the prefix 2^14 of the same vectors against the oracle
    """
    import pyoracle as po
    from helpers import pt_ints

    n = 1 << 22
    srs = ctx.srs_generate(0x1111, 0x2222, n)
    s1, s2 = rand_fr(n, 301), rand_fr(n, 302)
    d1, d2 = ctx.to_device(s1), ctx.to_device(s2)
    assert ctx.lib.zk_msm_window(n) == 19
    a = pt_ints(jac_norm_to_affine(ctx.msm_g1(srs, d1, n)))
    b = pt_ints(jac_norm_to_affine(ctx.msm_g1(srs, d2, n)))
    both = pt_ints(jac_norm_to_affine(ctx.msm_g1(srs, ctx.fr_add(d1, d2, n), n)))
    assert both == po.g1_add(a, b)
    m = 1 << 14
    bases = srs.download()[:m].copy()
    assert (jac_norm_to_affine(ctx.msm_g1(srs, d1, m)) == co.msm_g1(bases, s1[:m].copy())).all()


def test_device_buffers_are_recycled(ctx):
    """zk_free parks a block, zk_malloc of the same size hands it out again (no hipFree / device sync)"""
    a = ctx.alloc(12345 * 32)
    p = a.ptr
    a.free()
    b = ctx.alloc(12345 * 32)
    assert b.ptr == p
    c = ctx.alloc(12345 * 32)
    assert c.ptr != p


def test_async_jobs_match_blocking_calls(ctx, co):
    """zk_msm_g1_batch_async / zk_msm_wait: two jobs in flight with sumchecks running on the ctx stream in between;
    results bit-identical to the blocking batch (and through it to the oracle: test_batch_*), in any wait order"""
    sizes = [1 << 14, 300, 1 << 12, 0, 5000, 1 << 15]
    srs = [ctx.srs_generate(101 + i, 17 + 2 * i, max(n, 1)) for i, n in enumerate(sizes)]
    srs[0].precompute(0)  # one item on the window-table path
    sc = [rand_fr(max(n, 1), 900 + i) for i, n in enumerate(sizes)]
    dsc = [ctx.to_device(s) for s in sc]
    want = ctx.msm_g1_batch(srs, dsc, sizes)
    assert (jac_norm_to_affine(want[1]) == co.msm_g1(srs[1].download()[:300], sc[1][:300])).all()
    f, g, ch = rand_fr(1 << 16, 1), rand_fr(1 << 16, 2), rand_fr(16, 3)
    df, dg = ctx.to_device(f), ctx.to_device(g)
    tr0 = ctx.sumcheck_product(df, dg, 1 << 16, ch)
    j1 = ctx.msm_g1_batch_async(srs, dsc, sizes)
    tr1 = ctx.sumcheck_product(df, dg, 1 << 16, ch)  # shares the GPU (and would share scratch arenas) with the job
    j2 = ctx.msm_g1_batch_async(srs[:3], dsc[:3], sizes[:3])
    q, val = ctx.open_rounds(df, 1 << 16, ch)
    j3 = ctx.msm_g1_batch_async(srs, dsc, sizes)
    r2 = j2.wait()
    r1 = j1.wait()
    r3 = j3.wait()
    assert (r1 == want).all() and (r3 == want).all() and (r2 == want[:3]).all()
    assert (r1 is j1.wait())  # idempotent
    for a, b in zip(tr0, tr1):
        assert (a == b).all()
    assert (val == co.open_quotients(f, ch)[1]).all()
    # scalars produced on the ctx stream right before the job starts (the job must be ordered after them)
    a2 = ctx.fr_add(dsc[0], dsc[0], sizes[0])
    j = ctx.msm_g1_batch_async([srs[0]], [a2], [sizes[0]])
    got = jac_norm_to_affine(j.wait()[0])
    assert (got == co.g1_add_affine(jac_norm_to_affine(want[0]), jac_norm_to_affine(want[0]))).all()
    # an error at enqueue time leaves nothing behind
    import zkhip

    with pytest.raises(zkhip.MsmLengthError):
        ctx.msm_g1_batch_async([srs[1]], [dsc[0]], [sizes[0]])
    assert (ctx.msm_g1_batch_async(srs, dsc, sizes).wait() == want).all()


def test_arena_plan_round_trip_makes_the_first_pass_allocation_free():
    """
    zk_arena_plan_export / zk_arena_plan_import (include/zkhip.h): a ctx that imports the plan another ctx exported after a batch of
    MSMs and sumchecks runs the same work without growing an arena (export before == export after), with the same results; a
    buffer that is not a plan is refused.
    """
    import zkhip

    def work(c):
        n = 1 << 14
        srs = c.srs_generate(5, 7, n)
        srs.precompute(0)
        sc = c.to_device(rand_fr(n, 3))
        out = [c.msm_g1(srs, sc, n)]
        job = c.msm_g1_batch_async([srs, srs], [sc, sc], [n, n // 2])
        out.append(job.wait())
        f, g = c.to_device(rand_fr(1 << 15, 1)), c.to_device(rand_fr(1 << 15, 2))
        out.append(c.sumcheck_product(f, g, 1 << 15, rand_fr(15, 9))[0])
        return out

    a = zkhip.Ctx(0)
    ref = work(a)
    plan = a.arena_plan_export()
    assert plan[0] == 0x31304E414C504B5A and plan[2:].any()
    a.close()
    b = zkhip.Ctx(0)
    assert not b.arena_plan_export()[2:].any()
    b.arena_plan_import(plan)
    before = b.arena_plan_export()
    assert (before >= plan).all()
    got = work(b)
    assert (b.arena_plan_export() == before).all(), "the work grew an arena although the plan was imported"
    for x, y in zip(ref, got):
        assert (np.asarray(x) == np.asarray(y)).all()
    with pytest.raises(zkhip.ZkError):
        b.arena_plan_import(np.zeros(48, dtype=np.uint64))
    b.close()


def test_round6_sort_kernels_on_small_ragged_batches(ctx, co):
    """
    k_tab_hist / k_tab_scatter (level 1 straight from the scalars) and k_l2_* (level 2 in LDS-staged tiles) normally serve rows of
    >= 2^23 entries (test_gpu_bigsizes.py, test_gpu_fullsize.py); here the knobs force them onto small, ragged window-table batches --
    items of very different lengths in one class, offsets, table widths with and without a compile-time layout, skewed scalars -- and
    onto a table-less MSM (tiled level 2 only), against the C oracle.
    """
    n_max = 1 << 13
    bases, _ = synthetic_bases(n_max, 99)
    knobs = {"msm_fused_min": 10, "msm_l2_tiled": 64, "msm_np": 1024}
    try:
        for k, v in knobs.items():
            ctx.dbg_tune(k, v)
        srs = ctx.srs_register(bases)
        sc_tl = rand_fr(n_max, 5)
        assert (jac_norm_to_affine(ctx.msm_g1(srs, ctx.to_device(sc_tl), n_max)) == co.msm_g1(bases, sc_tl)).all()  # table-less, tiled level 2
        for width in (0, 13, 16, 19):  # 0: the library's pick (14 bits, run-time layout); 16, 19: compile-time layouts; 13: run-time
            srs.precompute(width)
            ns = [n_max, 5000, 1024, 777, 64, 3, 1]
            offs = [0, 100, 7000, 1, 8000, 8189, 8191]
            scs = [rand_fr(n, 300 + 10 * width + j) for j, n in enumerate(ns)]
            scs[1] = scs[1][np.zeros(ns[1], dtype=np.int64)]  # one scalar 5 000 times: a single bucket per window
            got = ctx.msm_g1_batch([srs] * len(ns), [ctx.to_device(s) for s in scs], ns, offsets=offs)
            for j, n in enumerate(ns):
                assert (jac_norm_to_affine(got[j]) == co.msm_g1(bases[offs[j] : offs[j] + n], scs[j])).all(), (width, n)
    finally:
        for k in knobs:
            ctx.dbg_tune(k, 0)
