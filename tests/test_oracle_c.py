"""The plain-C oracle (CPU baseline port) agrees bit-for-bit with the big-int oracle."""
import numpy as np
import pytest

import pyoracle as po
from helpers import pt_ints, pt_mont, synthetic_bases


def to_m(xs):
    return np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def from_m(a):
    return [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]


def test_rand_stream_and_field_ops(co):
    raw = co.rand_fr(11, 5)
    assert [sum(int(raw[i][k]) << (64 * k) for k in range(4)) for i in range(5)] == po.SplitMix64(11).fr_vec(5)
    rng = po.SplitMix64(1)
    xs, ys = rng.fr_vec(200) + [0, 1, po.R_MOD - 1], rng.fr_vec(200) + [po.R_MOD - 1, 0, po.R_MOD - 1]
    r = po.R_MOD
    assert from_m(co.fr_mul(to_m(xs), to_m(ys))) == [x * y % r for x, y in zip(xs, ys)]
    assert from_m(co.fr_add(to_m(xs), to_m(ys))) == [(x + y) % r for x, y in zip(xs, ys)]
    assert from_m(co.fr_sub(to_m(xs), to_m(ys))) == [(x - y) % r for x, y in zip(xs, ys)]
    nz = [y or 5 for y in ys]
    assert from_m(co.fr_div(to_m(xs), to_m(nz))) == [x * pow(y, -1, r) % r for x, y in zip(xs, nz)]
    with pytest.raises(ZeroDivisionError):
        co.fr_div(to_m([1]), to_m([0]))
    assert from_m(co.fr_to_mont(co.fr_from_mont(to_m(xs)))) == [x % r for x in xs]


@pytest.mark.parametrize("n", [1, 2, 6])
def test_multilinear_loops(co, n):
    rng = po.SplitMix64(100 + n)
    m = 1 << n
    f, g, ch = rng.fr_vec(m), rng.fr_vec(m), rng.fr_vec(n)
    assert [tuple(from_m(p)) for p in co.sumcheck(to_m(f), to_m(ch))] == po.sumcheck(f, ch)
    assert [tuple(from_m(p)) for p in co.sumcheck_product(to_m(f), to_m(g), to_m(ch))] == po.sumcheck_product(f, g, ch)
    assert from_m(co.fold(to_m(f), to_m(ch[:1]))) == po.fold(f, ch[0])
    q, v = co.open_quotients(to_m(f), to_m(ch))
    cur, qq = list(f), []
    for i in range(n):
        h = len(cur) // 2
        qq += [(cur[j + h] - cur[j]) % po.R_MOD for j in range(h)]
        cur = po.fold(cur, ch[i])
    assert from_m(q) == qq and from_m(v) == cur
    assert from_m(co.product_tree(to_m(f))) == po.product_tree(f)


def test_product_tree_reference_kat(co):
    t = from_m(co.product_tree(to_m([1, 2, 3, 4])))
    assert (t[0::2], t[1::2], t[4:]) == ([1, 3, 2, 24], [2, 4, 12, 0], [2, 12, 24, 0])


def test_g1_and_msm(co):
    rng = po.SplitMix64(9)
    k = rng.fr()
    kc = np.array([(k >> (64 * i)) & po.MASK64 for i in range(4)], dtype=np.uint64)
    assert pt_ints(co.g1_mul_affine(pt_mont(po.G1_GEN), kc)) == po.g1_mul(po.G1_GEN, k)
    for n in (1, 31, 32, 150):
        bases, _ = synthetic_bases(n, 3)
        assert [pt_ints(b) for b in bases] == po.g1_bases(n, 3)
        sc = rng.fr_vec(n)
        assert pt_ints(co.msm_g1(bases, to_m(sc))) == po.g1_msm([pt_ints(b) for b in bases], sc)
    # ark-ec window rule: 3 below 32 points, ceil_log2(n)*69/100 + 2 above
    assert co.msm_window(31) == 3 and co.msm_window(32) == 5 and co.msm_window(1 << 20) == 15
    # edge cases: zero / one / r-1 scalars, repeated and infinite bases, length mismatch
    b = [po.g1_mul(po.G1_GEN, 5)] * 3 + [None] + po.g1_bases(4, 8)
    s = [0, 1, po.R_MOD - 1, 5, 7, 0, po.R_MOD - 2, 3]
    assert pt_ints(co.msm_g1(np.array([pt_mont(P) for P in b]), to_m(s))) == po.g1_msm(b, s)
    with pytest.raises(ValueError):
        co.msm_g1(np.zeros((3, 12), np.uint64), to_m([1, 2]))
