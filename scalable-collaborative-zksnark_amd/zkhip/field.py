"""
numpy helpers for the reference's memory layouts (no arithmetic: that is the library's job).
Fr = [.., 4] uint64 Montgomery limbs; Fq = [.., 6]; affine G1 = [.., 12] (x||y, zeros = infinity).
"""
from __future__ import annotations

import numpy as np

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_M64 = (1 << 64) - 1
_FR_RINV = pow(1 << 256, -1, R_MOD)
_FQ_RINV = pow(1 << 384, -1, Q_MOD)
_R_LIMBS = np.array([(R_MOD >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def int_to_limbs(x: int, n: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(8 * n, "little"), dtype="<u8").astype(np.uint64)


def limbs_to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u8").tobytes(), "little")


def fr_sum_mont(parts) -> np.ndarray:
    """
    sum over axis 0 of Montgomery-form Fr arrays [P, ..., 4] -> [..., 4].  The Montgomery map is
    linear, so the raw values are added mod r without leaving Montgomery form.  Every party's array becomes ONE big
    integer with a 320-bit slot per element (64 spare bits: no carry crosses a slot for P < 2^64), the parties are
    added with P - 1 big-int additions, and each slot is reduced mod r on the way out.
    """
    a = np.stack([np.asarray(x, dtype=np.uint64) for x in parts])
    shape = a.shape[1:-1]
    a = a.reshape(a.shape[0], -1, 4)
    npart, m = a.shape[0], a.shape[1]
    wide = np.zeros((npart, m, 5), dtype="<u8")
    wide[:, :, :4] = a
    tot = 0
    for p in range(npart):
        tot += int.from_bytes(wide[p].tobytes(), "little")
    raw = tot.to_bytes(40 * m, "little")
    out = bytearray(32 * m)
    for j in range(m):
        out[32 * j : 32 * j + 32] = (int.from_bytes(raw[40 * j : 40 * j + 40], "little") % R_MOD).to_bytes(32, "little")
    return np.frombuffer(bytes(out), dtype="<u8").astype(np.uint64).reshape(*shape, 4)


def fr_mont(x: int) -> np.ndarray:
    """Montgomery limbs of the field element x (host big-int; for scalars like challenges)"""
    return int_to_limbs((x % R_MOD) * (1 << 256) % R_MOD, 4)


def fr_from_mont(a) -> int:
    return limbs_to_int(a) * _FR_RINV % R_MOD


def fq_mont(x: int) -> np.ndarray:
    return int_to_limbs((x % Q_MOD) * (1 << 384) % Q_MOD, 6)


def fq_from_mont(a) -> int:
    return limbs_to_int(a) * _FQ_RINV % Q_MOD


def random_fr(n: int, seed: int) -> np.ndarray:
    """
    n uniform field elements as [n,4] uint64 limbs < r (what `random_evaluations`,
    dist-primitive/src/lib.rs:13-18, produces: any canonical limb pattern is the Montgomery
    form of a uniform element).  Vectorised rejection sampling, seeded.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    while filled < n:
        m = max(16, int((n - filled) * 1.15) + 8)
        cand = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64)
        cand[:, 3] &= np.uint64(0x7FFFFFFFFFFFFFFF)
        # lexicographic compare with r from the top limb down
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for k in (3, 2, 1, 0):
            lt |= eq & (cand[:, k] < _R_LIMBS[k])
            eq &= cand[:, k] == _R_LIMBS[k]
        good = cand[lt]
        take = min(len(good), n - filled)
        out[filled : filled + take] = good[:take]
        filled += take
    return out


def splitmix_fr(n: int, seed: int) -> np.ndarray:
    """
    n uniform field elements from SplitMix64(seed) by rejection (SURVEY.md 8(d) "Synthetic inputs"): candidate k takes outputs
    4k+1 .. 4k+4 of the generator as limbs 0..3 (top bit of limb 3 cleared) and is kept when it is < r.  The generator's state is
    seed + k * gamma, so every output is computable on its own: vectorised here (cache-sized chunks, in-place arithmetic),
    sequential in the C++ host (host/zkhost/hyperplonk.hpp SplitMix64) -- the same elements.
    """
    gamma, m1, m2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)
    s30, s27, s31, top = np.uint64(30), np.uint64(27), np.uint64(31), np.uint64(0x7FFFFFFFFFFFFFFF)
    out = np.empty((n, 4), dtype=np.uint64)
    chunk = 1 << 16  # candidates per step: 2 MiB of state
    z, t = np.empty(4 * chunk, dtype=np.uint64), np.empty(4 * chunk, dtype=np.uint64)
    ramp = np.arange(1, 4 * chunk + 1, dtype=np.uint64) * gamma  # (k + 1) gamma for the outputs of one chunk
    filled, k0 = 0, 0  # candidates consumed so far
    r3, r2, r1, r0 = (_R_LIMBS[k] for k in (3, 2, 1, 0))
    with np.errstate(over="ignore"):
        while filled < n:
            np.add(ramp, np.uint64((seed + 4 * k0 * 0x9E3779B97F4A7C15) & _M64), out=z)
            np.right_shift(z, s30, out=t); np.bitwise_xor(z, t, out=z); np.multiply(z, m1, out=z)
            np.right_shift(z, s27, out=t); np.bitwise_xor(z, t, out=z); np.multiply(z, m2, out=z)
            np.right_shift(z, s31, out=t); np.bitwise_xor(z, t, out=z)
            cand = z.reshape(chunk, 4)
            cand[:, 3] &= top
            # < r, lexicographic from the top limb: almost every candidate is decided by limb 3 alone
            c3 = cand[:, 3]
            lt = c3 < r3
            eq = c3 == r3
            if eq.any():
                idx = np.nonzero(eq)[0]
                for i in idx:
                    lt[i] = (int(cand[i, 2]), int(cand[i, 1]), int(cand[i, 0])) < (int(r2), int(r1), int(r0))
            good = cand[lt]
            take = min(len(good), n - filled)
            out[filled : filled + take] = good[:take]
            filled += take
            k0 += chunk
    return out


def jacobian_to_affine_ints(j18):
    """normalised Jacobian (18 u64, as returned by zk_msm_g1) -> (x, y) ints or None"""
    j = np.asarray(j18, dtype=np.uint64).reshape(18)
    if not j[12:].any():
        return None
    return (fq_from_mont(j[0:6]), fq_from_mont(j[6:12]))


def affine_mont_to_ints(a12):
    a = np.asarray(a12, dtype=np.uint64).reshape(12)
    if not a.any():
        return None
    return (fq_from_mont(a[:6]), fq_from_mont(a[6:]))


def affine_ints_to_mont(P) -> np.ndarray:
    if P is None:
        return np.zeros(12, dtype=np.uint64)
    return np.concatenate([fq_mont(P[0]), fq_mont(P[1])])
