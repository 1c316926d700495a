"""GPU parity: sumcheck family, fold, open quotients, product tree vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from helpers import rand_fr

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 4, 64, 2048, 4096, 8192, 1 << 15, 1 << 17]


# 2^18 / 2^19: the single-table local stage whose first round runs straight out of the table (2^19), with and without an HBM pass
BIG1 = [1 << 18, 1 << 19]


@pytest.mark.parametrize("length", SIZES + BIG1)
def test_sumcheck(ctx, co, length):
    n = length.bit_length() - 1
    tab, chal = rand_fr(length, 100 + n), rand_fr(max(n, 1), 200 + n)
    pairs, last = ctx.sumcheck(ctx.to_device(tab), length, chal)
    exp = co.sumcheck(tab, chal)  # [n+1,2,4], last entry (0, last)
    assert (pairs == exp[:n]).all()
    assert (last == exp[n, 1]).all() and not exp[n, 0].any()


@pytest.mark.parametrize("length", SIZES + [1 << 18, 1 << 19, 1 << 21])  # from 2^18 the host derives t1 of the later rounds; 2^21: two passes with lazily reduced sums
def test_sumcheck_product(ctx, co, length):
    n = length.bit_length() - 1
    f, g, chal = rand_fr(length, 300 + n), rand_fr(length, 400 + n), rand_fr(max(n, 1), 500 + n)
    triples, lf, lg = ctx.sumcheck_product(ctx.to_device(f), ctx.to_device(g), length, chal)
    exp, elf, elg = co.sumcheck_product_rounds(f, g, chal)
    assert (triples == exp).all()
    assert (lf == elf).all() and (lg == elg).all()


@pytest.mark.parametrize("length,npts", [(1, 0), (8, 0), (8, 2), (8, 3), (8, 5), (4096, 2), (1 << 14, 2), (1 << 14, 14), (1 << 16, 5), (1 << 16, 9),
                                         (1 << 18, 18), (1 << 19, 1), (1 << 19, 3), (1 << 19, 19), (1 << 20, 2), (1 << 21, 12)])
def test_fold(ctx, co, length, npts):
    n = length.bit_length() - 1
    tab, pts = rand_fr(length, 600 + n), rand_fr(max(npts, 1), 700 + npts)[:npts]
    rounds = min(n, npts)
    got = ctx.fold(ctx.to_device(tab), length, pts).download((length >> rounds, 4))
    cur = tab
    for i in range(rounds):
        cur = co.fold(cur, pts[i])
    assert (got == cur).all()


@pytest.mark.parametrize("length", SIZES + BIG1)
def test_open_rounds(ctx, co, length):
    n = length.bit_length() - 1
    tab, pt = rand_fr(length, 800 + n), rand_fr(max(n, 1), 900 + n)
    q, val = ctx.open_rounds(ctx.to_device(tab), length, pt)
    eq, ev = co.open_quotients(tab, pt)
    assert (val == ev).all()
    if length > 1:
        assert (q.download((length - 1, 4)) == eq).all()


@pytest.mark.parametrize("N", [1, 2, 4, 8, 256, 512, 1024, 1 << 13, 1 << 16])
def test_product_tree(ctx, co, N):
    x = rand_fr(N, 1000 + N)
    got = ctx.product_tree(ctx.to_device(x), N).download((2 * N, 4))
    assert (got == co.product_tree(x)).all()


def test_product_tree_reference_kat(ctx, co):
    """dacc_product.rs:450-466: input [1,2,3,4] -> ([1,3,2,24],[2,4,12,0],[2,12,24,0])"""
    import pyoracle as po

    x = np.array([po.fr_to_mont_limbs(v) for v in (1, 2, 3, 4)], dtype=np.uint64)
    tree = ctx.product_tree(ctx.to_device(x), 4).download((8, 4))
    vals = [po.fr_from_mont_limbs(t) for t in tree]
    assert vals[0::2] == [1, 3, 2, 24] and vals[1::2] == [2, 4, 12, 0] and vals[4:] == [2, 12, 24, 0]


def test_batched_calls_match_single_calls(ctx, co):
    """zk_sumcheck_batch: a mixed batch (all four modes, sizes 2^0 .. 2^20 across every stage boundary, partial folds,
    repeated tables) gives the same bits as one call at a time, and as the oracle on the sampled items"""
    rng = np.random.default_rng(5)
    reqs, tabs = [], {}

    def tab(lg, seed):
        if (lg, seed) not in tabs:
            a = rand_fr(1 << lg, 7000 + 37 * lg + seed)
            tabs[(lg, seed)] = (a, ctx.to_device(a))
        return tabs[(lg, seed)]

    sizes = [0, 1, 2, 5, 8, 9, 10, 11, 13, 16, 17, 18, 19, 20, 18, 17, 9, 3]
    for i, lg in enumerate(sizes):
        ch = rand_fr(max(lg, 1) + 2, 8000 + i)
        kind = ("product", "plain", "fold", "open")[i % 4]
        if kind == "product":
            reqs.append(("product", tab(lg, 0)[1], tab(lg, 1)[1], 1 << lg, ch))
        elif kind == "plain":
            reqs.append(("plain", tab(lg, 0)[1], 1 << lg, ch))
        elif kind == "fold":
            reqs.append(("fold", tab(lg, 1)[1], 1 << lg, ch[: int(rng.integers(0, lg + 3))]))
        else:
            reqs.append(("open", tab(lg, 0)[1], 1 << lg, ch))
    # the layered shape of the wiring identity: three products per halving slice of the same tables
    base_f, base_g = tab(18, 0)[1], tab(18, 1)[1]
    off, clen = 0, 1 << 17
    while clen >= 1:
        ch = rand_fr(20, 9000 + clen.bit_length())
        reqs += [("product", base_f.at(32 * off), base_g.at(32 * off), clen, ch)] * 2
        off += clen // 2 if clen > 1 else 0
        clen //= 2
    got = ctx.sumcheck_batch(reqs)
    assert len(got) == len(reqs)
    for r, g in zip(reqs, got):
        if r[0] == "product":
            want = ctx.sumcheck_product(r[1], r[2], r[3], r[4])
            assert all((a == b).all() for a, b in zip(g, want))
        elif r[0] == "plain":
            want = ctx.sumcheck(r[1], r[2], r[3])
            assert all((a == b).all() for a, b in zip(g, want))
        elif r[0] == "fold":
            rounds = min(r[2].bit_length() - 1, len(r[3]))
            want = ctx.fold(r[1], r[2], r[3])
            assert (g.download((r[2] >> rounds, 4)) == want.download((r[2] >> rounds, 4))).all()
        else:
            qw, vw = ctx.open_rounds(r[1], r[2], r[3])
            assert (g[1] == vw).all()
            if r[2] > 1:
                assert (g[0].download((r[2] - 1, 4)) == qw.download((r[2] - 1, 4))).all()
    # oracle on two of them
    f20, g20 = tab(20, 0)[0], tab(20, 1)[0]
    i20 = sizes.index(20)
    assert reqs[i20][0] == "plain"
    exp = co.sumcheck(f20, reqs[i20][3][:20])
    assert (got[i20][0] == exp[:20]).all() and (got[i20][1] == exp[20, 1]).all()
    i19 = sizes.index(19)
    assert reqs[i19][0] == "product"
    etr, elf, elg = co.sumcheck_product_rounds(tab(19, 0)[0], tab(19, 1)[0], reqs[i19][4][:19])
    assert (got[i19][0] == etr).all() and (got[i19][1] == elf).all() and (got[i19][2] == elg).all()
    # errors name the item and leave the ctx usable
    import zkhip

    with pytest.raises(zkhip.ZkError):
        ctx.sumcheck_batch([("plain", tab(5, 0)[1], 32, rand_fr(5, 1)), ("plain", tab(5, 0)[1], 33, rand_fr(5, 1))])
    again = ctx.sumcheck_batch(reqs[:4])
    assert all((a == b).all() for a, b in zip(again[0], got[0]))
