"""
Oracle-backed stand-in for `zkhip.Ctx` used ONLY by the CPU tests of the host/exchange logic
(zkhip.dist_primitive over LocalTestNet / gloo): same method names, numpy arrays instead of HBM
buffers, arithmetic by the oracle.  It lives under tests/ -- the product never sees it.
"""
import numpy as np

import coracle as co
import pyoracle as po
from helpers import pt_ints, pt_mont


class NumpyBuf:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)

    def download(self, shape, dtype=np.uint64, offset=0):
        flat = self.a.reshape(-1)[offset // 8 :]
        n = int(np.prod(shape))
        return flat[:n].reshape(shape).copy()

    def at(self, byte_offset):
        return NumpyBuf(self.a.reshape(-1)[byte_offset // 8 :].reshape(-1, 4))


def _arr(x):
    return x.a if isinstance(x, NumpyBuf) else np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)


def _jac(aff12):
    out = np.zeros(18, dtype=np.uint64)
    a = np.asarray(aff12, dtype=np.uint64)
    one = np.array(po.fq_to_mont_limbs(1), dtype=np.uint64)
    if not a.any():
        out[0:6] = one
        out[6:12] = one
        return out
    out[:12] = a
    out[12:] = one
    return out


class OracleSrs:
    def __init__(self, bases):
        self.bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 12)

    def __len__(self):
        return len(self.bases)


class OracleBackend:
    def alloc(self, nbytes):
        return NumpyBuf(np.zeros((max(nbytes, 32) // 32, 4), dtype=np.uint64))

    def copy_d2d(self, dst, src, nbytes):
        _arr(dst).reshape(-1)[: nbytes // 8] = _arr(src).reshape(-1)[: nbytes // 8]

    def download_ptr(self, src, nbytes):
        return _arr(src).reshape(-1)[: nbytes // 8].copy().view(np.uint8)

    def upload_ptr(self, dst, a):
        a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        _arr(dst).reshape(-1)[: a.nbytes // 8] = a.view(np.uint64)

    def to_device(self, a):
        return NumpyBuf(a)

    def srs_register(self, bases, stride=96):
        return OracleSrs(bases)

    def msm_g1(self, srs, scalars, n, offset=0):
        return _jac(co.msm_g1(srs.bases[offset : offset + n], _arr(scalars)[:n]))

    def msm_g1_batch(self, srs_list, scalars_list, lens, offsets=None):
        out = np.zeros((len(lens), 18), dtype=np.uint64)
        for k, (s, sc, n) in enumerate(zip(srs_list, scalars_list, lens)):
            out[k] = self.msm_g1(s, sc, n, (offsets or [0] * len(lens))[k])
        return out

    def g1_lincomb(self, points, scalars_canon):
        acc = None
        for p, k in zip(np.asarray(points).reshape(-1, 18), np.asarray(scalars_canon).reshape(-1, 4)):
            P = None if not p[12:].any() else pt_ints(p[:12])
            kk = sum(int(k[i]) << (64 * i) for i in range(4))
            acc = po.g1_add(acc, po.g1_mul(P, kk))
        return _jac(pt_mont(acc))

    def g1_lincomb_batch(self, points, scalars_canon):
        pts = np.asarray(points, dtype=np.uint64)
        return np.stack([self.g1_lincomb(pts[r], scalars_canon) for r in range(pts.shape[0])]) if pts.shape[0] else np.zeros((0, 18), np.uint64)

    def sumcheck(self, tab, length, chal):
        r = co.sumcheck(_arr(tab)[:length], np.asarray(chal).reshape(-1, 4)) if length > 1 else None
        if length == 1:
            return np.zeros((0, 2, 4), np.uint64), _arr(tab)[0].copy()
        n = length.bit_length() - 1
        return r[:n], r[n, 1]

    def sumcheck_product(self, f, g, length, chal):
        if length == 1:
            return np.zeros((0, 3, 4), np.uint64), _arr(f)[0].copy(), _arr(g)[0].copy()
        return co.sumcheck_product_rounds(_arr(f)[:length], _arr(g)[:length], np.asarray(chal).reshape(-1, 4))

    def product_tree(self, x, N):
        return NumpyBuf(co.product_tree(_arr(x)[:N]))

    def open_rounds(self, tab, length, point, q_out=None):
        q, v = co.open_quotients(_arr(tab)[:length], np.asarray(point).reshape(-1, 4))
        if q_out is not None:  # (a caller-provided quotient buffer, like the library's q_out)
            _arr(q_out)[: len(q)] = q
            return q_out, v
        return NumpyBuf(q if len(q) else np.zeros((1, 4), np.uint64)), v

    # element-wise / table helpers used by the protocol driver
    def fr_add(self, a, b, n, out=None):
        return NumpyBuf(co.fr_add(_arr(a)[:n], _arr(b)[:n]))

    def fr_sub(self, a, b, n, out=None):
        return NumpyBuf(co.fr_sub(_arr(a)[:n], _arr(b)[:n]))

    def fr_mul(self, a, b, n, out=None):
        return NumpyBuf(co.fr_mul(_arr(a)[:n], _arr(b)[:n]))

    def fr_axpb(self, a, b, alpha, beta, n, out=None):
        al = np.tile(np.asarray(alpha, dtype=np.uint64).reshape(1, 4), (n, 1))
        be = np.tile(np.asarray(beta, dtype=np.uint64).reshape(1, 4), (n, 1))
        return NumpyBuf(co.fr_add(co.fr_add(_arr(a)[:n], co.fr_mul(al, _arr(b)[:n])), be))

    def fr_scale(self, b, alpha, n, out=None):
        al = np.tile(np.asarray(alpha, dtype=np.uint64).reshape(1, 4), (n, 1))
        return NumpyBuf(co.fr_mul(al, _arr(b)[:n]))

    def fr_batch_div(self, num, den, n, out=None):
        return NumpyBuf(co.fr_div(_arr(num)[:n], _arr(den)[:n]))

    def fr_apply_matrix(self, matrix, d_in, in_vec_stride, in_comp_stride, k, out_vec_stride, out_row_stride, out=None):
        m = np.asarray(matrix, dtype=np.uint64)
        rows, cols = m.shape[0], m.shape[1]
        src = _arr(d_in)
        span = (k - 1) * out_vec_stride + (rows - 1) * out_row_stride + 1 if k and rows else 1
        res = np.zeros((span, 4), dtype=np.uint64)
        for r in range(rows):
            acc = np.zeros((k, 4), dtype=np.uint64)
            for c in range(cols):
                col = src[c * in_comp_stride : c * in_comp_stride + (k - 1) * in_vec_stride + 1 : in_vec_stride]
                acc = co.fr_add(acc, co.fr_mul(np.tile(m[r, c].reshape(1, 4), (k, 1)), np.ascontiguousarray(col)))
            res[r * out_row_stride : r * out_row_stride + (k - 1) * out_vec_stride + 1 : out_vec_stride] = acc
        return NumpyBuf(res)

    def fr_ntt_map(self, tables, d_in, in_vec_stride, in_comp_stride, k, out_vec_stride, out_row_stride, out=None):
        """
        zk_fr_ntt_map's contract restated with python big-ints and DENSE sums (no butterfly code shared with the kernel): per
        vector j, coefficients c_i = sum_x in[x] winv^(x i) over the n_in inputs (domain size A), y_i = scale_i c_i for
        i < min(A, B) and 0 above, outputs out[r] = sum_i y_i w^(i r step) for r < take.  The branch of dist_primitive taken
        from 64 parties up (pss.rs:93-171 as the reference itself runs it: ifft, resize, fft).
        """
        t = tables
        A, B, nin, take, step = t["A"], t["B"], t["n_in"], t["take"], t["step"]
        dec = lambda a: [po.fr_from_mont_limbs(r) for r in np.asarray(a, dtype=np.uint64).reshape(-1, 4)]
        root = lambda tab, size: 1 if size == 1 else (po.R_MOD - 1 if size == 2 else tab[1])
        wi, w, scale = root(dec(t["winv"]), A), root(dec(t["w"]), B), dec(t["scale"])
        wip = [pow(wi, e, po.R_MOD) for e in range(A)]
        wp = [pow(w, e, po.R_MOD) for e in range(B)]
        keep = min(A, B)
        src = _arr(d_in)
        span = (k - 1) * out_vec_stride + (take - 1) * out_row_stride + 1 if k else 1
        res = np.zeros((span, 4), dtype=np.uint64)
        for j in range(k):
            x = [po.fr_from_mont_limbs(src[j * in_vec_stride + c * in_comp_stride]) for c in range(nin)]
            y = [scale[i] * sum(x[c] * wip[(c * i) % A] for c in range(nin)) % po.R_MOD for i in range(keep)]
            for r in range(take):
                e = r * step
                res[j * out_vec_stride + r * out_row_stride] = po.fr_to_mont_limbs(sum(y[i] * wp[(i * e) % B] for i in range(keep)) % po.R_MOD)
        return NumpyBuf(res)

    def fr_deinterleave(self, t, n):
        a = _arr(t)[: 2 * n]
        return NumpyBuf(a[0::2].copy()), NumpyBuf(a[1::2].copy())

    def fold(self, tab, length, points, out=None):
        cur = _arr(tab)[:length]
        pts = np.asarray(points, dtype=np.uint64).reshape(-1, 4)
        for i in range(min(length.bit_length() - 1, len(pts))):
            cur = co.fold(cur, pts[i])
        return NumpyBuf(cur)

    def srs_generate(self, k0, k1, n):
        from helpers import pt_mont
        start = pt_mont(po.g1_mul(po.G1_GEN, k0))
        step = pt_mont(po.g1_mul(po.G1_GEN, k1))
        return OracleSrs(co.g1_arith_seq(start, step, n))
