"""
Packed secret sharing parameters -- host-side mirror of secret-sharing/src/pss.rs:17-172.

Every PSS map is a fixed PUBLIC linear map over Fr on vectors of N_p = 8l entries, so it is
kept as small integer matrices (python ints) and applied either to Fr values on the host or,
for points, through the library's K9 entry point (zk_g1_lincomb).  The ark-poly semantics the
reference relies on are isolated in `_fft` / `_ifft`: `fft_in_place` / `ifft_in_place` first
resize (zero-pad or truncate) the vector to the domain size (SURVEY.md Appendix C -- an
assumption that cannot be checked against arkworks in this environment).
"""
from __future__ import annotations

from typing import List, Sequence

from .field import R_MOD

FR_GENERATOR = 7
TWO_ADICITY = 32
ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> TWO_ADICITY, R_MOD)


class _Domain:
    def __init__(self, size: int, offset: int = 1):
        assert size & (size - 1) == 0
        self.size, self.offset = size, offset % R_MOD
        self.omega = pow(ROOT_OF_UNITY, (1 << TWO_ADICITY) // size, R_MOD)

    def _resize(self, v: Sequence[int]) -> List[int]:
        v = list(v[: self.size])
        return v + [0] * (self.size - len(v))

    def _pows(self):
        if not hasattr(self, "_w"):
            w, winv = [1], [1]
            wi = pow(self.omega, -1, R_MOD)
            for _ in range(self.size - 1):
                w.append(w[-1] * self.omega % R_MOD)
                winv.append(winv[-1] * wi % R_MOD)
            self._w, self._winv = w, winv
        return self._w, self._winv

    def _transform(self, v: List[int], roots: List[int]) -> List[int]:
        """in-place radix-2 transform: out[j] = sum_k v[k] * root^(jk) -- the same values as the sums of the definition (exact
        arithmetic mod r), n log n multiplications instead of n^2 (256 parties: 60 x fewer; the constructor maps every unit vector)"""
        n = self.size
        j = 0
        for i in range(1, n):  # bit reversal
            bit = n >> 1
            while j & bit:
                j ^= bit
                bit >>= 1
            j ^= bit
            if i < j:
                v[i], v[j] = v[j], v[i]
        ln = 2
        while ln <= n:
            half, step = ln >> 1, n // ln
            for i in range(0, n, ln):
                for k in range(half):
                    u, t = v[i + k], v[i + k + half] * roots[k * step] % R_MOD
                    v[i + k] = (u + t) % R_MOD
                    v[i + k + half] = (u - t) % R_MOD
            ln <<= 1
        return v

    def fft(self, coeffs: Sequence[int]) -> List[int]:
        """evaluate sum_k c_k x^k at x_j = offset * omega^j"""
        c = self._resize(coeffs)
        w, _ = self._pows()
        cs, op = [], 1
        for k in range(self.size):  # fold the coset offset into the coefficients
            cs.append(c[k] * op % R_MOD)
            op = op * self.offset % R_MOD
        return self._transform(cs, w)

    def ifft(self, evals: Sequence[int]) -> List[int]:
        e = [x % R_MOD for x in self._resize(evals)]
        _, winv = self._pows()
        self._transform(e, winv)
        op, oinv = pow(self.size, -1, R_MOD), pow(self.offset, -1, R_MOD)
        out = []
        for k in range(self.size):
            out.append(e[k] * op % R_MOD)
            op = op * oinv % R_MOD
        return out


class PackedSharingParams:
    """PackedSharingParams::new(l) (pss.rs:38-64): n = 8l parties, t = l-1"""

    def __init__(self, l: int):
        assert l >= 1 and l & (l - 1) == 0
        self.l, self.n, self.t = l, 8 * l, l - 1
        self.share = _Domain(self.n)
        self.secret = _Domain(2 * l, FR_GENERATOR)
        self.secret2 = _Domain(4 * l, FR_GENERATOR)
        unit = lambda m, j: [1 if k == j else 0 for k in range(m)]
        tr = lambda cols: [list(r) for r in zip(*cols)]
        # share_i = sum_j pack[i][j] * secret_j        (secrets zero-padded to 2l)
        self.pack_matrix = tr([self.pack_from_public(unit(2 * l, j)) for j in range(2 * l)])
        # secret_j = sum_i unpack[j][i] * share_i
        self.unpack_matrix = tr([self.unpack(unit(self.n, i)) for i in range(self.n)])
        self.unpack2_matrix = tr([self.unpack2(unit(self.n, i)) for i in range(self.n)])

    # --- the reference's maps on Fr vectors (python ints) ---
    def pack_from_public(self, secrets: Sequence[int]) -> List[int]:
        """pss.rs:69-73,93-99"""
        return self.share.fft(self.secret.ifft(secrets))

    def pack_single(self, secret: int) -> List[int]:
        """pss.rs:103-113 -- packs and then packs the n-vector AGAIN (reference quirk, kept literally)"""
        return self.pack_from_public(self.share.fft(self.secret.ifft([secret])))

    def pack_single_of_one(self) -> List[int]:
        """pack_single is linear in its one argument: pack_single(s)[p] = s * pack_single(1)[p]; pss2ss (unpack.rs:72-97) needs entry p
        of pack_single of every unpacked secret -- one multiplication each instead of four transforms"""
        if not hasattr(self, "_single_one"):
            self._single_one = self.pack_single(1)
        return self._single_one

    def unpack(self, shares: Sequence[int]) -> List[int]:
        """pss.rs:117-120,132-149"""
        return self.secret.fft(self.share.ifft(shares))[: self.l]

    def unpack2(self, shares: Sequence[int]) -> List[int]:
        """pss.rs:124-128,153-171 (slots 0,2,..,2l-2)"""
        assert len(shares) == self.n
        return self.secret2.fft(self.share.ifft(shares))[0 : 2 * self.l : 2]

    def ntt_tables(self, kind: str) -> dict:
        """
        twiddles and scales of one map for zk_fr_ntt_map (Montgomery limbs): the transform form of the reference's
        own FFT pipeline -- ifft on the input domain, resize, fft on the output domain (pss.rs:93-171).
        kind: "pack" (l secrets -> 8l shares), "unpack" (8l -> l), "unpack2" (8l -> l, slots 0, 2, ..).
        """
        import numpy as np

        from .field import fr_mont

        cache = self.__dict__.setdefault("_ntt", {})
        if kind in cache:
            return cache[kind]
        l = self.l
        src, dst, n_in, take, step = {"pack": (self.secret, self.share, l, self.n, 1), "unpack": (self.share, self.secret, self.n, l, 1),
                                      "unpack2": (self.share, self.secret2, self.n, l, 2)}[kind]
        A, B = src.size, dst.size
        winv = pow(src.omega, -1, R_MOD)
        ratio = dst.offset * pow(src.offset, -1, R_MOD) % R_MOD
        ainv = pow(A, -1, R_MOD)
        lim = lambda xs: np.array([fr_mont(x) for x in xs], dtype=np.uint64).reshape(-1, 4)
        t = dict(A=A, B=B, n_in=n_in, take=take, step=step,
                 winv=lim([pow(winv, i, R_MOD) for i in range(max(A // 2, 1))]),
                 w=lim([pow(dst.omega, i, R_MOD) for i in range(max(B // 2, 1))]),
                 scale=lim([ainv * pow(ratio, i, R_MOD) % R_MOD for i in range(min(A, B))]))
        cache[kind] = t
        return t

    # --- coefficient rows used by the distributed primitives ---
    def dmsm_coeffs(self, party: int) -> List[int]:
        """
        d_msm leader closure (dmsm.rs:30-39) for one output party p:
            out_p = pack_from_public([S; l])[p],  S = sum_j unpack2(shares)_j
                  = sum_i (c_p * lambda_i) * C_i
        returns [c_p * lambda_i mod r for i < n].
        """
        lam = [sum(self.unpack2_matrix[j][i] for j in range(self.l)) % R_MOD for i in range(self.n)]
        c_p = sum(self.pack_matrix[party][j] for j in range(self.l)) % R_MOD
        return [c_p * x % R_MOD for x in lam]
