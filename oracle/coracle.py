"""
ctypes front-end for oracle/libzkoracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY -- see the header of zk_oracle.c.  Arrays are numpy
uint64 in the reference's memory layout (Montgomery limbs, little-endian).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzkoracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "zk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libzkoracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _binary(name, width):
    def f(a, b):
        a, b = _u64(a), _u64(b)
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(b), _p(out), ctypes.c_size_t(a.size // width))
        return out

    return f


fr_mul = _binary("ora_fr_mul", 4)
fr_add = _binary("ora_fr_add", 4)
fr_sub = _binary("ora_fr_sub", 4)
fq_mul = _binary("ora_fq_mul", 6)
fq_add = _binary("ora_fq_add", 6)
fq_sub = _binary("ora_fq_sub", 6)


def _unary(name, width):
    def f(a):
        a = _u64(a)
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(out), ctypes.c_size_t(a.size // width))
        return out

    return f


fr_to_mont = _unary("ora_fr_to_mont", 4)
fr_from_mont = _unary("ora_fr_from_mont", 4)
fq_to_mont = _unary("ora_fq_to_mont", 6)
fq_from_mont = _unary("ora_fq_from_mont", 6)


def fr_div(a, b):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    rc = lib().ora_fr_div(_p(a), _p(b), _p(out), ctypes.c_size_t(a.size // 4))
    if rc != 0:
        raise ZeroDivisionError("zero denominator")
    return out


def rand_fr(seed: int, n: int) -> np.ndarray:
    """n uniform Fr as raw limbs [n,4] (same stream as pyoracle.SplitMix64(seed).fr())"""
    out = np.empty((n, 4), dtype=np.uint64)
    lib().ora_rand_fr(ctypes.c_uint64(seed), _p(out), ctypes.c_size_t(n))
    return out


def fold(tab, r):
    tab, r = _u64(tab), _u64(r)
    m = tab.size // 4
    out = np.empty((m // 2, 4), dtype=np.uint64)
    lib().ora_fold(_p(tab), ctypes.c_size_t(m), _p(r), _p(out))
    return out


def sumcheck(tab, chal):
    """-> [n+1, 2, 4]"""
    tab, chal = _u64(tab), _u64(chal)
    m = tab.size // 4
    n = m.bit_length() - 1
    out = np.empty((n + 1, 2, 4), dtype=np.uint64)
    lib().ora_sumcheck(_p(tab), ctypes.c_size_t(m), _p(chal), _p(out))
    return out


def sumcheck_product(f, g, chal):
    """-> [n+1, 3, 4]"""
    f, g, chal = _u64(f), _u64(g), _u64(chal)
    m = f.size // 4
    n = m.bit_length() - 1
    out = np.empty((n + 1, 3, 4), dtype=np.uint64)
    lib().ora_sumcheck_product(_p(f), _p(g), ctypes.c_size_t(m), _p(chal), _p(out))
    return out


def sumcheck_product_rounds(f, g, chal):
    """-> (triples [n,3,4], last_f [4], last_g [4])"""
    f, g, chal = _u64(f), _u64(g), _u64(chal)
    m = f.size // 4
    n = m.bit_length() - 1
    out = np.empty((n, 3, 4), dtype=np.uint64)
    lf = np.empty(4, dtype=np.uint64)
    lg = np.empty(4, dtype=np.uint64)
    lib().ora_sumcheck_product_rounds(_p(f), _p(g), ctypes.c_size_t(m), _p(chal), _p(out), _p(lf), _p(lg))
    return out, lf, lg


def open_quotients(tab, point):
    """-> (q [len-1, 4] concatenated q_0 | q_1 | ..., value [4])"""
    tab, point = _u64(tab), _u64(point)
    m = tab.size // 4
    q = np.empty((max(m - 1, 0), 4), dtype=np.uint64)
    v = np.empty(4, dtype=np.uint64)
    lib().ora_open_quotients(_p(tab), ctypes.c_size_t(m), _p(point), _p(q), _p(v))
    return q, v


def product_tree(x):
    x = _u64(x)
    n = x.size // 4
    out = np.empty((2 * n, 4), dtype=np.uint64)
    lib().ora_product_tree(_p(x), ctypes.c_size_t(n), _p(out))
    return out


def msm_window(n: int) -> int:
    return lib().ora_msm_window(ctypes.c_size_t(n))


def msm_g1(bases, scalars):
    """bases [n,12] affine Montgomery (x=y=0: infinity), scalars [n,4] Montgomery -> affine [12]"""
    bases, scalars = _u64(bases), _u64(scalars)
    n = scalars.size // 4
    if bases.size // 12 != n:
        raise ValueError(min(bases.size // 12, n))
    out = np.empty(12, dtype=np.uint64)
    lib().ora_msm_g1(_p(bases), _p(scalars), ctypes.c_size_t(n), _p(out))
    return out


def g1_add_affine(p, q):
    p, q = _u64(p), _u64(q)
    out = np.empty(12, dtype=np.uint64)
    lib().ora_g1_add_affine(_p(p), _p(q), _p(out))
    return out


def g1_mul_affine(p, k_canon):
    p, k = _u64(p), _u64(k_canon)
    out = np.empty(12, dtype=np.uint64)
    lib().ora_g1_mul_affine(_p(p), _p(k), _p(out))
    return out


def g1_jac_to_affine(jac):
    jac = _u64(jac)
    out = np.empty(12, dtype=np.uint64)
    lib().ora_g1_jac_to_affine(_p(jac), _p(out))
    return out


def g1_arith_seq(start_aff, step_aff, n):
    """P_i = start + i*step, i < n, affine Montgomery [n,12]"""
    s, t = _u64(start_aff), _u64(step_aff)
    out = np.empty((n, 12), dtype=np.uint64)
    lib().ora_g1_arith_seq(_p(s), _p(t), ctypes.c_size_t(n), _p(out))
    return out


