"""
An INDEPENDENT statement of the packed-secret-sharing maps (secret-sharing/src/pss.rs:93-171) by their
mathematical definition -- polynomial interpolation with explicit coefficient vectors (Newton / Horner on
python ints), no DFT matrices and none of the domain code shared by zkhip.pss._Domain and
pyoracle.Radix2Domain -- checked against both implementations at l = 1, 2, 4, 8.

  pack_from_public(v)  p of degree < 2l with p(g w_2l^j) = v_j (v zero-padded to 2l); share_i = p(w_8l^i)
  unpack(s)            p of degree < 8l through (w_8l^i, s_i); keep the first 2l COEFFICIENTS (what ark-poly's
                       fft_in_place does to a longer vector: resize = truncate); secrets = p'(g w_2l^j), j < l
  unpack2(s)           same with 4l coefficients on the coset of size 4l; slots 0, 2, .., 2l-2

(What cannot be checked in this image is the `resize` behaviour of ark-poly itself: SURVEY.md Appendix C.)
"""
import random

import pytest

import pyoracle as po
from zkhip.pss import PackedSharingParams

R = po.R_MOD
G = 7
ROOT = pow(G, (R - 1) >> 32, R)  # 2^32-th root of unity: TWO_ADIC_ROOT_OF_UNITY of ark-bls12-381 Fr


def root(k):
    return pow(ROOT, (1 << 32) // k, R)


def interpolate(xs, ys):
    """coefficients (lowest first) of the unique polynomial of degree < len(xs) through (xs, ys): Newton form"""
    n = len(xs)
    dd = list(ys)
    for j in range(1, n):  # divided differences
        for i in range(n - 1, j - 1, -1):
            dd[i] = (dd[i] - dd[i - 1]) * pow(xs[i] - xs[i - j], -1, R) % R
    coeffs = [0] * n
    basis = [1] + [0] * (n - 1)  # prod (x - xs[k]), k < j
    for j in range(n):
        for k in range(n):
            coeffs[k] = (coeffs[k] + dd[j] * basis[k]) % R
        nb = [0] * n
        for k in range(n - 1):
            nb[k + 1] = basis[k]
        for k in range(n):
            nb[k] = (nb[k] - xs[j] * basis[k]) % R
        basis = nb
    return coeffs


def horner(c, x):
    acc = 0
    for a in reversed(c):
        acc = (acc * x + a) % R
    return acc


def def_pack(v, l):
    v = list(v) + [0] * (2 * l - len(v))
    xs = [G * pow(root(2 * l), j, R) % R for j in range(2 * l)]
    c = interpolate(xs, v)
    return [horner(c, pow(root(8 * l), i, R)) for i in range(8 * l)]


def def_unpack(s, l, two=False):
    xs = [pow(root(8 * l), i, R) for i in range(8 * l)]
    c = interpolate(xs, s)
    k = 4 * l if two else 2 * l
    c = c[:k]  # the truncation of fft_in_place on a longer vector
    ev = [horner(c, G * pow(root(k), j, R) % R) for j in range(k)]
    return ev[0 : 2 * l : 2] if two else ev[:l]


@pytest.mark.parametrize("l", [1, 2, 4, 8])
def test_pss_maps_against_the_definition(l):
    rng = random.Random(100 + l)
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    for trial in range(2):
        v = [rng.randrange(R) for _ in range(l if trial == 0 else 2 * l)]
        exp = def_pack(v, l)
        assert pp.pack_from_public(v) == exp and opp.pack_from_public(v) == exp
        s = [rng.randrange(R) for _ in range(8 * l)]  # arbitrary shares: full degree, the truncation matters
        assert pp.unpack(s) == def_unpack(s, l) == opp.unpack(s)
        assert pp.unpack2(s) == def_unpack(s, l, two=True) == opp.unpack2(s)
    # the matrices the device applies are these maps
    for i in range(8 * l):
        unit = [1 if k == i else 0 for k in range(8 * l)]
        assert [row[i] for row in pp.unpack_matrix] == def_unpack(unit, l)
        assert [row[i] for row in pp.unpack2_matrix] == def_unpack(unit, l, two=True)
    for j in range(2 * l):
        unit = [1 if k == j else 0 for k in range(2 * l)]
        assert [row[j] for row in pp.pack_matrix] == def_pack(unit, l)


@pytest.mark.parametrize("l", [2, 4])
def test_pack_single_quirk_and_roundtrips(l):
    """pack_single packs twice (pss.rs:103-113); pack -> unpack and products of shares -> unpack2 round-trip"""
    rng = random.Random(7 + l)
    pp = PackedSharingParams(l)
    x = rng.randrange(R)
    assert pp.pack_single(x) == def_pack(def_pack([x], l)[: 2 * l], l)
    a, b = [rng.randrange(R) for _ in range(l)], [rng.randrange(R) for _ in range(l)]
    sa, sb = def_pack(a, l), def_pack(b, l)
    assert def_unpack(sa, l) == a
    prod = [x * y % R for x, y in zip(sa, sb)]  # degree < 4l: unpack2 recovers the element-wise products
    assert pp.unpack2(prod) == [x * y % R for x, y in zip(a, b)]
