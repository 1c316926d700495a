"""shared test utilities: seeded inputs in the reference's memory layouts"""
import numpy as np

import coracle as co
import pyoracle as po


def pt_mont(P):
    if P is None:
        return np.zeros(12, dtype=np.uint64)
    return np.array(po.fq_to_mont_limbs(P[0]) + po.fq_to_mont_limbs(P[1]), dtype=np.uint64)


def pt_ints(a12):
    a = np.asarray(a12, dtype=np.uint64).reshape(12)
    if not a.any():
        return None
    return (po.fq_from_mont_limbs(a[:6]), po.fq_from_mont_limbs(a[6:]))


def jac_norm_to_affine(j18):
    """normalised Jacobian [18] (library output) -> affine Montgomery [12] (zeros = infinity)"""
    j = np.asarray(j18, dtype=np.uint64).reshape(18)
    if not j[12:].any():
        return np.zeros(12, dtype=np.uint64)
    one = np.array(po.fq_to_mont_limbs(1), dtype=np.uint64)
    assert (j[12:] == one).all(), "library results must be normalised (z = R mod q)"
    return j[:12].copy()


_G = None


def synthetic_bases(n: int, seed: int) -> np.ndarray:
    """P_i = (k0 + i*k1)*G as affine Montgomery [n,12] -- same points as pyoracle.g1_bases(n, seed)"""
    rng = po.SplitMix64(seed)
    k0, k1 = rng.fr(), rng.fr()
    start = pt_mont(po.g1_mul(po.G1_GEN, k0))
    step = pt_mont(po.g1_mul(po.G1_GEN, k1))
    return co.g1_arith_seq(start, step, n), (k0, k1)


def rand_fr(n: int, seed: int) -> np.ndarray:
    """uniform Fr limbs [n,4] (interpreted as Montgomery form)"""
    return co.rand_fr(seed, n)


def oracle_msm_chunked(bases: np.ndarray, scalars: np.ndarray, threads: int = 0) -> np.ndarray:
    """
    the C oracle's MSM on many host threads: contiguous chunks (ctypes releases the GIL), partial results added with
    the python oracle's group law -> affine Montgomery [12].  A 2^24-point MSM that takes ~200 s on one core of the
    port finishes in seconds on the GPU box's cores.
    """
    import os
    from concurrent.futures import ThreadPoolExecutor

    m = len(scalars)
    threads = threads or max(1, min(64, (os.cpu_count() or 1) // 2))
    chunks = max(threads, (m + (1 << 18) - 1) >> 18)  # <= 2^18 points per chunk keeps every worker busy to the end
    bounds = [m * i // chunks for i in range(chunks + 1)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda i: co.msm_g1(bases[bounds[i] : bounds[i + 1]], scalars[bounds[i] : bounds[i + 1]]), range(chunks)))
    acc = None
    for pt in parts:
        acc = po.g1_add(acc, pt_ints(pt))
    return pt_mont(acc)
