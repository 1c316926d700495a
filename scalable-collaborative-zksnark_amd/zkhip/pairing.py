"""
A BLS12-381 pairing on host big-ints, for `PolynomialCommitment::verify` (dist-primitive/src/dpoly_comm.rs:466-484)
-- the reference's only end-to-end soundness check (`should_commit_and_open`, :502-531).  It is OFF the hot path:
the verifier runs once per opened polynomial on a handful of points, so plain Python integers are enough
(a pairing product with n + 1 Miller loops and ONE final exponentiation takes seconds).

Construction: Fq12 = Fq[w] / (w^12 - 2 w^6 + 2) (so u = w^6 - 1 satisfies u^2 = -1 and Fq2 = Fq[u] embeds), the
G2 point is moved from the twist y^2 = x^3 + 4 (1 + u) to E(Fq12): y^2 = x^3 + 4 by (x, y) -> (x / w^2, y / w^3),
and the Miller loop runs over |x| = 0xd201000000010000 with generic affine line functions on E(Fq12).  The map is
bilinear and non-degenerate on G1 x G2 (`tests/test_pairing.py` checks e(aP, bQ) = e(P, Q)^(ab) like the
reference's `should_pair`, :495-500); `verify` only compares products of pairings, so any such map serves.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from .field import Q_MOD, R_MOD

Q = Q_MOD
ATE_LOOP = 0xD201000000010000  # |x|, the BLS12-381 parameter
DEG = 12
# w^12 = 2 w^6 - 2
_MOD_LOW = (2, 6), (-2, 0)  # (coefficient, exponent) of the reduction of w^12


class Fq12:
    """element of Fq[w] / (w^12 - 2 w^6 + 2) as 12 coefficients (lowest first)"""

    __slots__ = ("c",)

    def __init__(self, c: Sequence[int]):
        self.c = [x % Q for x in c]

    @staticmethod
    def one() -> "Fq12":
        return Fq12([1] + [0] * 11)

    @staticmethod
    def zero() -> "Fq12":
        return Fq12([0] * 12)

    @staticmethod
    def from_fq(x: int) -> "Fq12":
        return Fq12([x] + [0] * 11)

    @staticmethod
    def from_fq2(a: Tuple[int, int]) -> "Fq12":
        """a0 + a1 u with u = w^6 - 1"""
        return Fq12([a[0] - a[1], 0, 0, 0, 0, 0, a[1], 0, 0, 0, 0, 0])

    def __eq__(self, o) -> bool:
        return self.c == o.c

    def __add__(self, o: "Fq12") -> "Fq12":
        return Fq12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o: "Fq12") -> "Fq12":
        return Fq12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self) -> "Fq12":
        return Fq12([-a for a in self.c])

    def scale(self, k: int) -> "Fq12":
        return Fq12([a * k for a in self.c])

    def __mul__(self, o: "Fq12") -> "Fq12":
        t = [0] * 23
        a, b = self.c, o.c
        for i in range(12):
            ai = a[i]
            if ai:
                for j in range(12):
                    t[i + j] += ai * b[j]
        for k in range(22, 11, -1):  # w^k = 2 w^(k-6) - 2 w^(k-12)
            v = t[k]
            if v:
                t[k - 6] += 2 * v
                t[k - 12] -= 2 * v
        return Fq12(t[:12])

    def is_zero(self) -> bool:
        return not any(self.c)

    def inv(self) -> "Fq12":
        """extended Euclid on polynomials over Fq against the modulus w^12 - 2 w^6 + 2"""
        def deg(p):
            d = len(p) - 1
            while d >= 0 and p[d] % Q == 0:
                d -= 1
            return d

        def poly_divmod(num, den):
            num = [x % Q for x in num]
            dd = deg(den)
            inv_lead = pow(den[dd], -1, Q)
            out = [0] * max(1, len(num) - dd)
            for i in range(deg(num) - dd, -1, -1):
                coef = num[i + dd] * inv_lead % Q
                out[i] = coef
                if coef:
                    for j in range(dd + 1):
                        num[i + j] = (num[i + j] - coef * den[j]) % Q
            return out, num[:dd] if dd > 0 else [0]

        modulus = [2, 0, 0, 0, 0, 0, -2 % Q, 0, 0, 0, 0, 0, 1]
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = list(self.c) + [0], list(modulus)
        while deg(low) > 0:
            qt, _ = poly_divmod(high, low)
            qt += [0] * (13 - len(qt))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] -= lm[i] * qt[j]
                    new[i + j] -= low[i] * qt[j]
            nm = [x % Q for x in nm]
            new = [x % Q for x in new]
            lm, low, hm, high = nm, new, lm, low
        k = pow(low[0], -1, Q)
        return Fq12([x * k for x in lm[:12]])

    def __pow__(self, e: int) -> "Fq12":
        r, b = Fq12.one(), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r


W = Fq12([0, 1] + [0] * 10)
_W2_INV = (W * W).inv()
_W3_INV = (W * W * W).inv()

P12 = Optional[Tuple[Fq12, Fq12]]


def _embed_g1(P) -> P12:
    return None if P is None else (Fq12.from_fq(P[0]), Fq12.from_fq(P[1]))


def _untwist(Qp) -> P12:
    """G2 point on the twist -> E(Fq12)"""
    if Qp is None:
        return None
    return (Fq12.from_fq2(Qp[0]) * _W2_INV, Fq12.from_fq2(Qp[1]) * _W3_INV)


def _double(P: P12) -> P12:
    x, y = P
    lam = (x * x).scale(3) * (y.scale(2)).inv()
    nx = lam * lam - x.scale(2)
    return (nx, lam * (x - nx) - y)


def _add(P: P12, R: P12) -> P12:
    if P is None:
        return R
    if R is None:
        return P
    if P[0] == R[0]:
        return _double(P) if P[1] == R[1] else None
    lam = (R[1] - P[1]) * (R[0] - P[0]).inv()
    nx = lam * lam - P[0] - R[0]
    return (nx, lam * (P[0] - nx) - P[1])


def _line(P1: P12, P2: P12, T: P12) -> Fq12:
    """the line through P1 and P2 (tangent if equal) evaluated at T"""
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if not (x1 == x2):
        lam = (y2 - y1) * (x2 - x1).inv()
        return lam * (xt - x1) - (yt - y1)
    if y1 == y2:
        lam = (x1 * x1).scale(3) * (y1.scale(2)).inv()
        return lam * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(Qp, P) -> Fq12:
    """f_{|x|, Q}(P) for Q in G2 (twist coordinates, ((x0, x1), (y0, y1))) and P in G1 ((x, y)); 1 if either is infinity"""
    if Qp is None or P is None:
        return Fq12.one()
    Q12, P12_ = _untwist(Qp), _embed_g1(P)
    R, f = Q12, Fq12.one()
    for bit in bin(ATE_LOOP)[3:]:
        f = f * f * _line(R, R, P12_)
        R = _double(R)
        if bit == "1":
            f = f * _line(R, Q12, P12_)
            R = _add(R, Q12)
    return f


def final_exponentiation(f: Fq12) -> Fq12:
    return f ** ((Q**12 - 1) // R_MOD)


def pairing(Qp, P) -> Fq12:
    return final_exponentiation(miller_loop(Qp, P))


def pairing_product_is_one(pairs: Sequence[Tuple[object, object]]) -> bool:
    """prod_i e(P_i, Q_i) == 1 with ONE final exponentiation; pairs = [(P in G1, Q in G2), ...]"""
    f = Fq12.one()
    for P, Qp in pairs:
        f = f * miller_loop(Qp, P)
    return final_exponentiation(f) == Fq12.one()


# ---- the little G1 / G2 affine arithmetic the verifier needs (host, python ints) ----
def _fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def _fq2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * n % Q, -a[1] * n % Q)


def g2_add(P, R):
    if P is None:
        return R
    if R is None:
        return P
    (x1, y1), (x2, y2) = P, R
    if x1 == x2:
        if y1 != y2 or y1 == (0, 0):
            return None
        lam = _fq2_mul(_fq2_mul((3, 0), _fq2_mul(x1, x1)), _fq2_inv(((2 * y1[0]) % Q, (2 * y1[1]) % Q)))
    else:
        lam = _fq2_mul(((y2[0] - y1[0]) % Q, (y2[1] - y1[1]) % Q), _fq2_inv(((x2[0] - x1[0]) % Q, (x2[1] - x1[1]) % Q)))
    l2 = _fq2_mul(lam, lam)
    x3 = ((l2[0] - x1[0] - x2[0]) % Q, (l2[1] - x1[1] - x2[1]) % Q)
    t = _fq2_mul(lam, ((x1[0] - x3[0]) % Q, (x1[1] - x3[1]) % Q))
    return (x3, ((t[0] - y1[0]) % Q, (t[1] - y1[1]) % Q))


def g2_neg(P):
    return None if P is None else (P[0], ((-P[1][0]) % Q, (-P[1][1]) % Q))


def g2_mul(P, k: int):
    k %= R_MOD
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, P)
    return acc


def g1_add(P, R):
    if P is None:
        return R
    if R is None:
        return P
    (x1, y1), (x2, y2) = P, R
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % Q)


def g1_mul(P, k: int):
    k %= R_MOD
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, P)
    return acc


G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)


def powers_of_g2(s: Sequence[int], g2=G2_GEN) -> List:
    """[g2, g2 * s_0, g2 * s_1, ...] (PolynomialCommitmentCub::new, dpoly_comm.rs:59-62)"""
    return [g2] + [g2_mul(g2, x) for x in s]


def verify(g1, powers_g2: Sequence, commitment, value: int, proof: Sequence, point: Sequence[int]) -> bool:
    """
    PolynomialCommitment::verify (dpoly_comm.rs:466-484):
        e(C - value * g1, g2) == sum_i e(proof_i, powers_of_g2[i + 1] - point_i * g2)
    as ONE pairing product: e(-(C - value g1), g2) * prod_i e(proof_i, s_i g2 - point_i g2) == 1.
    Points are affine python-int tuples (None = infinity), value / point canonical integers.
    """
    g2 = powers_g2[0]
    left = g1_add(commitment, g1_neg(g1_mul(g1, value)))
    pairs = [(g1_neg(left), g2)]
    for i, pi in enumerate(proof):
        pairs.append((pi, g2_add(powers_g2[i + 1], g2_neg(g2_mul(g2, point[i])))))
    return pairing_product_is_one(pairs)
