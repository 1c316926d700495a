"""
Thin object wrapper over the C ABI: `Ctx` = one GPU (zk_ctx), `DeviceBuffer`, `Srs`.
Arguments that name device memory accept a DeviceBuffer, a torch CUDA tensor (data_ptr) or a
raw integer address.  Host-side results come back as numpy uint64 limb arrays.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib
from ._lib import ZK_ERR_DIV_ZERO, ZK_ERR_LENGTH, test_hooks


_TRACE_MSM = bool(os.environ.get("ZKHIP_TRACE_MSM"))


class ZkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"zkhip error {code}: {msg}")
        self.code = code


class MsmLengthError(ZkError):
    """`G::msm` -> Err(min_len) (ark-ec 0.4.2), which dmsm.rs:23 unwrap()s"""

    def __init__(self, code, msg, min_len):
        super().__init__(code, msg)
        self.min_len = min_len


def _ptr(x) -> int:
    if x is None:
        return 0
    if isinstance(x, (DeviceBuffer, DeviceView)):
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):  # torch tensor
        return int(x.data_ptr())
    raise TypeError(f"not a device buffer: {type(x)}")


def _h(a: np.ndarray) -> int:
    return a.ctypes.data


class DeviceBuffer:
    def __init__(self, ctx: "Ctx", nbytes: int):
        self.ctx = ctx
        self.nbytes = nbytes
        p = ctypes.c_void_p()
        ctx._check(ctx.lib.zk_malloc(ctx.h, nbytes, ctypes.byref(p)))
        self.ptr = p.value or 0

    def free(self):
        if self.ptr:
            self.ctx.lib.zk_free(self.ctx.h, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.zk_memcpy_h2d(self.ctx.h, self.ptr + offset, _h(a), a.nbytes))
        return self

    def download(self, shape, dtype=np.uint64, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.zk_memcpy_d2h(self.ctx.h, _h(out), self.ptr + offset, out.nbytes))
        return out

    def at(self, byte_offset: int) -> "DeviceView":
        """a view of this buffer from byte_offset on.  The view REFERENCES its parent: whoever holds the view (a queued MSM
        item, a traced sumcheck operand) keeps the allocation alive, so its address cannot be handed to another buffer."""
        return DeviceView(self, byte_offset)


class DeviceView:
    """an address inside a DeviceBuffer that keeps the buffer alive; accepted wherever a device buffer is"""

    __slots__ = ("parent", "offset", "ptr")

    def __init__(self, parent: DeviceBuffer, byte_offset: int):
        assert 0 <= byte_offset <= parent.nbytes, "view outside its buffer"
        self.parent, self.offset, self.ptr = parent, byte_offset, parent.ptr + byte_offset

    @property
    def ctx(self):
        return self.parent.ctx

    @property
    def nbytes(self) -> int:
        return self.parent.nbytes - self.offset

    def at(self, byte_offset: int) -> "DeviceView":
        return DeviceView(self.parent, self.offset + byte_offset)

    def download(self, shape, dtype=np.uint64, offset: int = 0) -> np.ndarray:
        return self.parent.download(shape, dtype, self.offset + offset)

    def upload(self, a: np.ndarray, offset: int = 0):
        self.parent.upload(a, self.offset + offset)
        return self

    def __int__(self) -> int:
        return self.ptr


class Srs:
    def __init__(self, ctx: "Ctx", handle: int):
        self.ctx, self.h = ctx, handle

    def __len__(self):
        return self.ctx.lib.zk_srs_len(self.h)

    @property
    def device_ptr(self) -> int:
        return self.ctx.lib.zk_srs_device_ptr(self.h) or 0

    def precompute(self, window_bits: int = 0, record_bytes: int = 0):
        """build the per-window tables (setup): MSMs on this SRS then share one bucket set.  record_bytes (G1): 96 packed (the default),
        128 = one record per 128-byte line (faster gathers, 4/3 of the table memory)"""
        if record_bytes:
            self.ctx._check(self.ctx.lib.zk_srs_precompute_layout(self.ctx.h, self.h, window_bits, record_bytes))
        else:
            self.ctx._check(self.ctx.lib.zk_srs_precompute(self.ctx.h, self.h, window_bits))
        return self

    @property
    def table_record(self) -> int:
        """bytes per record of the precomputed table, 0 if none"""
        return self.ctx.lib.zk_srs_table_record(self.h)

    @property
    def table_window(self) -> int:
        """window bits of the precomputed table, 0 if none"""
        return self.ctx.lib.zk_srs_table_window(self.h)

    def download(self) -> np.ndarray:
        n = len(self)
        out = np.empty((n, 24 if getattr(self, "g2", False) else 12), dtype=np.uint64)
        if n:
            self.ctx._check(self.ctx.lib.zk_srs_download(self.ctx.h, self.h, _h(out)))
        return out

    def free(self):
        if self.h:
            self.ctx.lib.zk_srs_free(self.ctx.h, self.h)
            self.h = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def comm_init_all(ctxs) -> None:
    """zk_comm_init_all: one process holding a ctx per GPU (the reference's one-task-per-party model,
    mpc-net/src/multi.rs:330-352); ctxs[p] becomes party p.  Drive every party from its own thread afterwards."""
    lib = _lib.lib()
    arr = (ctypes.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    rc = lib.zk_comm_init_all(arr, len(ctxs))
    if rc != 0:
        raise ZkError(rc, (lib.zk_last_error(ctxs[0].h) or b"").decode() if ctxs else "zk_comm_init_all")


class MsmJob:
    """an asynchronous batch of MSMs in flight (zk_msm_g1_batch_async / zk_msm_wait)"""

    def __init__(self, ctx: "Ctx", srs_list, scalars_list, lens, offsets=None):
        self.ctx, self.count = ctx, len(lens)
        self.keep = (list(srs_list), list(scalars_list))  # inputs must outlive the job
        self.lens = [int(x) for x in lens]
        self.h = None
        self.res = None
        if self.count == 0:
            self.res = np.zeros((0, 18), dtype=np.uint64)
            return
        count = self.count
        h = (ctypes.c_void_p * count)(*[s.h for s in srs_list])
        sp = (ctypes.c_void_p * count)(*[_ptr(s) for s in scalars_list])
        nn = (ctypes.c_size_t * count)(*self.lens)
        off = (ctypes.c_size_t * count)(*[int(x) for x in (offsets or [0] * count)])
        job = ctypes.c_void_p()
        rc = ctx.lib.zk_msm_g1_batch_async(ctx.h, count, h, off, sp, nn, ctypes.byref(job))
        if rc == ZK_ERR_LENGTH:
            raise MsmLengthError(rc, (ctx.lib.zk_last_error(ctx.h) or b"").decode(), 0)
        ctx._check(rc)
        self.h = job.value

    def wait(self) -> np.ndarray:
        if self.res is None:
            out = np.zeros((self.count, 18), dtype=np.uint64)
            h, self.h = self.h, None
            self.ctx._check(self.ctx.lib.zk_msm_wait(self.ctx.h, h, _h(out)))
            self.res, self.keep = out, None
        return self.res

    def __del__(self):
        try:
            if self.h and getattr(self.ctx, "h", None):  # never waited for: drain and release it
                out = np.zeros((self.count, 18), dtype=np.uint64)
                self.ctx.lib.zk_msm_wait(self.ctx.h, self.h, _h(out))
                self.h = None
        except Exception:
            pass


class Ctx:
    def __init__(self, device: int = 0):
        self.lib = _lib.lib()
        h = ctypes.c_void_p()
        rc = self.lib.zk_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise ZkError(rc, "zk_ctx_create failed (no MI355X visible?) -- zkhip has no CPU fallback")
        self.h = h.value
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.zk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise ZkError(rc, (self.lib.zk_last_error(self.h) or b"").decode())

    def set_stream(self, hip_stream: int):
        self._check(self.lib.zk_ctx_set_stream(self.h, hip_stream))

    def sync(self):
        self._check(self.lib.zk_ctx_sync(self.h))

    # ---- memory ----
    def mem_info(self):
        """(free, total) bytes of this ctx's GPU"""
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self.lib.zk_mem_info(self.h, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def arena_plan_export(self) -> np.ndarray:
        """the sizes this ctx's scratch arenas have grown to (zk_arena_plan_export): 48 u64 -- keep them beside the proving key"""
        plan = np.zeros(48, dtype=np.uint64)
        self._check(self.lib.zk_arena_plan_export(self.h, _h(plan)))
        return plan

    def arena_plan_import(self, plan):
        """grow the arenas to a plan exported after an earlier proof of the same shape: the first proof then allocates nothing"""
        plan = np.ascontiguousarray(plan, dtype=np.uint64).reshape(-1)
        assert plan.size == 48
        self._check(self.lib.zk_arena_plan_import(self.h, _h(plan)))

    def trim(self) -> int:
        """hand every parked zk_free block back to the driver; returns the bytes released"""
        f = ctypes.c_size_t(0)
        self._check(self.lib.zk_trim(self.h, ctypes.byref(f)))
        return f.value

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def temp(self, nbytes: int, key) -> DeviceBuffer:
        """a cached device buffer of at least nbytes for `key` (avoids hipMalloc/hipFree per call)"""
        cache = self.__dict__.setdefault("_temps", {})
        buf = cache.get(key)
        if buf is None or buf.nbytes < nbytes:
            buf = cache[key] = DeviceBuffer(self, max(nbytes, 1))
        return buf

    def fr_scale(self, b, alpha: np.ndarray, n: int, out=None):
        """out = alpha * b (element-wise, one public scalar)"""
        out = out or self.alloc(max(32 * n, 1))
        al = np.ascontiguousarray(alpha, dtype=np.uint64)
        zero = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.zk_fr_axpb(self.h, 0, _ptr(b), _h(al), _h(zero), _ptr(out), n))
        return out

    def copy_d2d(self, dst, src, nbytes: int):
        self._check(self.lib.zk_memcpy_d2d(self.h, _ptr(dst), _ptr(src), nbytes))

    def download_ptr(self, src, nbytes: int) -> np.ndarray:
        """nbytes from a device buffer / raw device address -> uint8 array"""
        out = np.empty(nbytes, dtype=np.uint8)
        if nbytes:
            self._check(self.lib.zk_memcpy_d2h(self.h, _h(out), _ptr(src), nbytes))
        return out

    def upload_ptr(self, dst, a: np.ndarray):
        a = np.ascontiguousarray(a)
        if a.nbytes:
            self._check(self.lib.zk_memcpy_h2d(self.h, _ptr(dst), _h(a), a.nbytes))

    def to_device(self, a: np.ndarray) -> DeviceBuffer:
        a = np.ascontiguousarray(a)
        return DeviceBuffer(self, max(a.nbytes, 1)).upload(a) if a.nbytes else DeviceBuffer(self, 1)

    # ---- element-wise ----
    def _binary(self, fn, a, b, n, out=None) -> DeviceBuffer:
        out = out or self.alloc(max(32 * n, 1))
        self._check(fn(self.h, _ptr(a), _ptr(b), _ptr(out), n))
        return out

    def fr_add(self, a, b, n, out=None):
        return self._binary(self.lib.zk_fr_add, a, b, n, out)

    def fr_sub(self, a, b, n, out=None):
        return self._binary(self.lib.zk_fr_sub, a, b, n, out)

    def fr_mul(self, a, b, n, out=None):
        return self._binary(self.lib.zk_fr_mul, a, b, n, out)

    def fr_apply_matrix(self, matrix: np.ndarray, d_in, in_vec_stride: int, in_comp_stride: int, k: int, out_vec_stride: int, out_row_stride: int, out=None):
        """out[j*osv + r*osr] = sum_c M[r][c] * in[j*isv + c*isc]; matrix [rows, cols, 4] Montgomery limbs"""
        m = np.ascontiguousarray(matrix, dtype=np.uint64)
        rows, cols = m.shape[0], m.shape[1]
        span = (k - 1) * out_vec_stride + (rows - 1) * out_row_stride + 1 if k and rows else 1
        out = out or self.alloc(32 * span)
        self._check(self.lib.zk_fr_apply_matrix(self.h, _h(m), rows, cols, _ptr(d_in), in_vec_stride, in_comp_stride, _ptr(out), out_vec_stride, out_row_stride, k))
        return out

    def fr_ntt_map(self, tables: dict, d_in, in_vec_stride: int, in_comp_stride: int, k: int, out_vec_stride: int, out_row_stride: int, out=None):
        """a PSS map by transforms (zk_fr_ntt_map); tables = PackedSharingParams.ntt_tables(kind)"""
        t = tables
        span = (k - 1) * out_vec_stride + (t["take"] - 1) * out_row_stride + 1 if k else 1
        out = out or self.alloc(32 * span)
        self._check(self.lib.zk_fr_ntt_map(self.h, t["A"], _h(t["winv"]), t["B"], _h(t["w"]), _h(t["scale"]), t["n_in"], t["take"], t["step"],
                                           _ptr(d_in), in_vec_stride, in_comp_stride, _ptr(out), out_vec_stride, out_row_stride, k))
        return out

    def fr_deinterleave(self, t, n):
        """(t[0::2], t[1::2]) for a table of 2n Fr -> two device buffers of n Fr"""
        even, odd = self.alloc(max(32 * n, 1)), self.alloc(max(32 * n, 1))
        self._check(self.lib.zk_fr_deinterleave(self.h, _ptr(t), _ptr(even), _ptr(odd), n))
        return even, odd

    def fr_batch_div(self, num, den, n, out=None):
        out = out or self.alloc(max(32 * n, 1))
        rc = self.lib.zk_fr_batch_div(self.h, _ptr(num), _ptr(den), _ptr(out), n)
        if rc == ZK_ERR_DIV_ZERO:
            raise ZeroDivisionError("zero denominator (the reference panics on inverse().unwrap())")
        self._check(rc)
        return out

    def fr_axpb(self, a, b, alpha: np.ndarray, beta: np.ndarray, n, out=None):
        out = out or self.alloc(max(32 * n, 1))
        al, be = np.ascontiguousarray(alpha, dtype=np.uint64), np.ascontiguousarray(beta, dtype=np.uint64)
        self._check(self.lib.zk_fr_axpb(self.h, _ptr(a), _ptr(b), _h(al), _h(be), _ptr(out), n))
        return out

    # ---- sumcheck family ----
    def sumcheck(self, tab, length: int, chal: np.ndarray):
        """-> (pairs [n,2,4], last [4])"""
        n = length.bit_length() - 1
        chal = np.ascontiguousarray(chal, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((n, 2, 4), dtype=np.uint64)
        last = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.zk_sumcheck(self.h, _ptr(tab), length, _h(chal), _h(out), _h(last)))
        return out, last

    def sumcheck_product(self, f, g, length: int, chal: np.ndarray):
        """-> (triples [n,3,4], last_f [4], last_g [4])"""
        n = length.bit_length() - 1
        chal = np.ascontiguousarray(chal, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((n, 3, 4), dtype=np.uint64)
        lf = np.zeros(4, dtype=np.uint64)
        lg = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.zk_sumcheck_product(self.h, _ptr(f), _ptr(g), length, _h(chal), _h(out), _h(lf), _h(lg)))
        return out, lf, lg

    def sumcheck_batch(self, reqs):
        """
        several INDEPENDENT calls of the family in one C-ABI call (zk_sumcheck_batch): one enqueue over several streams, one
        completion.  reqs: list of
            ("plain", tab, length, chal) | ("product", f, g, length, chal) | ("fold", tab, length, points[, out]) | ("open", tab, length, point[, q_out])
        -> list of what the single calls return: (pairs, last) | (triples, last_f, last_g) | folded buffer | (q buffer, value).
        """
        from ._lib import ScItem

        n = len(reqs)
        if n == 0:
            return []
        items = (ScItem * n)()
        keep, outs = [], []
        for i, r in enumerate(reqs):
            kind = r[0]
            it = items[i]
            if kind == "product":
                _, f, g, length, chal = r
                k = length.bit_length() - 1
                ch = np.ascontiguousarray(chal, dtype=np.uint64).reshape(-1, 4)[:k]
                tr, lf, lg = np.zeros((k, 3, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
                it.mode, it.d_f, it.d_g, it.len, it.h_chal, it.h_sums, it.h_last_f, it.h_last_g = 1, _ptr(f), _ptr(g), length, _h(ch), _h(tr), _h(lf), _h(lg)
                outs.append((tr, lf, lg))
            elif kind == "plain":
                _, f, length, chal = r
                k = length.bit_length() - 1
                ch = np.ascontiguousarray(chal, dtype=np.uint64).reshape(-1, 4)[:k]
                pr, last = np.zeros((k, 2, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64)
                it.mode, it.d_f, it.len, it.h_chal, it.h_sums, it.h_last_f = 0, _ptr(f), length, _h(ch), _h(pr), _h(last)
                outs.append((pr, last))
            elif kind == "fold":
                _, f, length, points = r[:4]
                k = length.bit_length() - 1
                ch = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 4)
                out = (r[4] if len(r) > 4 else None) or self.alloc(32 * (length >> min(k, len(ch))))
                it.mode, it.d_f, it.len, it.h_chal, it.n_points, it.d_out = 2, _ptr(f), length, _h(ch), len(ch), _ptr(out)
                outs.append(out)
            elif kind == "open":
                _, f, length, point = r[:4]
                k = length.bit_length() - 1
                ch = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)[:k]
                q = (r[4] if len(r) > 4 else None) or self.alloc(max(32 * (length - 1), 1))
                val = np.zeros(4, dtype=np.uint64)
                it.mode, it.d_f, it.len, it.h_chal, it.h_last_f, it.d_out = 3, _ptr(f), length, _h(ch), _h(val), _ptr(q)
                outs.append((q, val))
            else:
                raise ValueError(kind)
            keep.append(ch)
        self._check(self.lib.zk_sumcheck_batch(self.h, n, items))
        return outs

    def fold(self, tab, length: int, points: np.ndarray, out=None):
        points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 4)
        n = length.bit_length() - 1
        rounds = min(n, len(points))
        out = out or self.alloc(32 * (length >> rounds))
        self._check(self.lib.zk_fold(self.h, _ptr(tab), length, _h(points), len(points), _ptr(out)))
        return out

    def open_rounds(self, tab, length: int, point: np.ndarray, q_out=None):
        """-> (q device buffer with length-1 Fr, value [4])"""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        q_out = q_out or self.alloc(max(32 * (length - 1), 1))
        val = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.zk_open_rounds(self.h, _ptr(tab), length, _h(point), _ptr(q_out), _h(val)))
        return q_out, val

    def product_tree(self, x, N: int, out=None):
        out = out or self.alloc(64 * N)
        self._check(self.lib.zk_product_tree(self.h, _ptr(x), N, _ptr(out)))
        return out

    # ---- MSM ----
    def srs_register(self, bases: np.ndarray, stride: int = 96) -> Srs:
        b = np.ascontiguousarray(bases)
        n = b.nbytes // stride
        h = ctypes.c_void_p()
        self._check(self.lib.zk_srs_register(self.h, _h(b), stride, n, ctypes.byref(h)))
        return Srs(self, h.value)

    def srs_wrap_device(self, d_bases, n: int) -> Srs:
        h = ctypes.c_void_p()
        self._check(self.lib.zk_srs_wrap_device(self.h, _ptr(d_bases), n, ctypes.byref(h)))
        return Srs(self, h.value)

    def srs_generate(self, k0: int, k1: int, n: int) -> Srs:
        from .field import int_to_limbs

        a, b = int_to_limbs(k0, 4), int_to_limbs(k1, 4)
        h = ctypes.c_void_p()
        self._check(self.lib.zk_srs_generate(self.h, _h(a), _h(b), n, ctypes.byref(h)))
        return Srs(self, h.value)

    def srs_powers(self, s: np.ndarray, g: np.ndarray = None):
        """PolynomialCommitmentCub::new (dpoly_comm.rs:37-67): s [n,4] Montgomery -> list of n+1 Srs levels"""
        s = np.ascontiguousarray(s, dtype=np.uint64).reshape(-1, 4)
        n = len(s)
        handles = (ctypes.c_void_p * (n + 1))()
        gp = np.ascontiguousarray(g, dtype=np.uint64) if g is not None else None
        self._check(self.lib.zk_srs_powers(self.h, _h(gp) if gp is not None else 0, _h(s) if n else 0, n, handles))
        return [Srs(self, handles[k]) for k in range(n + 1)]

    def srs_to_packed(self, level: Srs, row_canon: np.ndarray, l: int) -> Srs:
        """to_packed (dpoly_comm.rs:164-194) for one party: row_canon [l,4] canonical pack coefficients"""
        row = np.ascontiguousarray(row_canon, dtype=np.uint64).reshape(-1, 4)
        assert len(row) >= min(l, len(level))
        h = ctypes.c_void_p()
        self._check(self.lib.zk_srs_to_packed(self.h, level.h, _h(row), l, ctypes.byref(h)))
        return Srs(self, h.value)

    def g1_apply_matrix(self, matrix_canon: np.ndarray, d_in, in_vec_stride: int, in_comp_stride: int, k: int, out_vec_stride: int, out_row_stride: int, out=None):
        """zk_fr_apply_matrix on affine G1 points (reference layout, 96 B): matrix [rows, cols, 4] CANONICAL scalars"""
        m = np.ascontiguousarray(matrix_canon, dtype=np.uint64)
        rows, cols = m.shape[0], m.shape[1]
        span = (k - 1) * out_vec_stride + (rows - 1) * out_row_stride + 1 if k and rows else 1
        out = out or self.alloc(96 * span)
        self._check(self.lib.zk_g1_apply_matrix(self.h, _h(m), rows, cols, _ptr(d_in), in_vec_stride, in_comp_stride, _ptr(out), out_vec_stride, out_row_stride, k))
        return out

    def msm_g1(self, srs: Srs, scalars, n: int, offset: int = 0) -> np.ndarray:
        """-> normalised Jacobian [18] uint64"""
        out = np.zeros(18, dtype=np.uint64)
        rc = self.lib.zk_msm_g1(self.h, srs.h, offset, _ptr(scalars), n, _h(out))
        if rc == ZK_ERR_LENGTH:
            raise MsmLengthError(rc, (self.lib.zk_last_error(self.h) or b"").decode(), min(n, max(len(srs) - offset, 0)))
        self._check(rc)
        if _TRACE_MSM:
            import sys

            print("zkhip-msm", [int(n)], [round(float(x), 3) for x in self.msm_last_timing()], file=sys.stderr)
        return out

    def msm_g1_batch(self, srs_list, scalars_list, lens, offsets=None) -> np.ndarray:
        """a batch of independent MSMs in one pipeline pass -> [count, 18]"""
        count = len(lens)
        out = np.zeros((count, 18), dtype=np.uint64)
        if count == 0:
            return out
        h = (ctypes.c_void_p * count)(*[s.h for s in srs_list])
        sp = (ctypes.c_void_p * count)(*[_ptr(s) for s in scalars_list])
        nn = (ctypes.c_size_t * count)(*[int(x) for x in lens])
        off = (ctypes.c_size_t * count)(*[int(x) for x in (offsets or [0] * count)])
        rc = self.lib.zk_msm_g1_batch(self.h, count, h, off, sp, nn, _h(out))
        if rc == ZK_ERR_LENGTH:
            raise MsmLengthError(rc, (self.lib.zk_last_error(self.h) or b"").decode(), 0)
        self._check(rc)
        if _TRACE_MSM:  # diagnostics: one line per batch (sizes + the library's phase timers)
            import sys

            print("zkhip-msm", [int(x) for x in lens], [round(float(x), 3) for x in self.msm_last_timing()], file=sys.stderr)
        return out

    def msm_g1_batch_async(self, srs_list, scalars_list, lens, offsets=None):
        """start a batch of MSMs on the ctx's job streams; -> job with .wait() -> [count, 18].  The scalar buffers must
        stay alive and unmodified until wait() (the job object keeps references to them)."""
        return MsmJob(self, srs_list, scalars_list, lens, offsets)

    # ---- G2 (same pipeline over Fq2) ----
    def srs_register_g2(self, bases: np.ndarray, stride: int = 192) -> Srs:
        """bases: affine G2 points x.c0|x.c1|y.c0|y.c1, [n, 24] uint64 Montgomery limbs (zeros = infinity)"""
        b = np.ascontiguousarray(bases)
        n = b.nbytes // stride
        h = ctypes.c_void_p()
        self._check(self.lib.zk_srs_register_g2(self.h, _h(b), stride, n, ctypes.byref(h)))
        s = Srs(self, h.value)
        s.g2 = True
        return s

    def msm_g2(self, srs: Srs, scalars, n: int, offset: int = 0) -> np.ndarray:
        """-> normalised Jacobian over Fq2, [36] uint64 (x.c0 x.c1 y.c0 y.c1 z.c0 z.c1)"""
        out = np.zeros(36, dtype=np.uint64)
        rc = self.lib.zk_msm_g2(self.h, srs.h, offset, _ptr(scalars), n, _h(out))
        if rc == ZK_ERR_LENGTH:
            raise MsmLengthError(rc, (self.lib.zk_last_error(self.h) or b"").decode(), min(n, max(len(srs) - offset, 0)))
        self._check(rc)
        return out

    def msm_g2_batch(self, srs_list, scalars_list, lens, offsets=None) -> np.ndarray:
        count = len(lens)
        out = np.zeros((count, 36), dtype=np.uint64)
        if count == 0:
            return out
        h = (ctypes.c_void_p * count)(*[s.h for s in srs_list])
        sp = (ctypes.c_void_p * count)(*[_ptr(s) for s in scalars_list])
        nn = (ctypes.c_size_t * count)(*[int(x) for x in lens])
        off = (ctypes.c_size_t * count)(*[int(x) for x in (offsets or [0] * count)])
        self._check(self.lib.zk_msm_g2_batch(self.h, count, h, off, sp, nn, _h(out)))
        return out

    def msm_g1_host(self, bases: np.ndarray, scalars: np.ndarray, stride: int = 96) -> np.ndarray:
        """drop-in for G::msm(&[Affine], &[Fr]) on host arrays"""
        b = np.ascontiguousarray(bases)
        s = np.ascontiguousarray(scalars, dtype=np.uint64)
        nb, ns = b.nbytes // stride, s.size // 4
        out = np.zeros(18, dtype=np.uint64)
        err = ctypes.c_size_t(0)
        rc = self.lib.zk_msm_g1_host(self.h, _h(b), stride, nb, _h(s), ns, _h(out), ctypes.byref(err))
        if rc == ZK_ERR_LENGTH:
            raise MsmLengthError(rc, (self.lib.zk_last_error(self.h) or b"").decode(), err.value)
        self._check(rc)
        return out

    def g1_lincomb(self, points: np.ndarray, scalars_canon: np.ndarray) -> np.ndarray:
        """sum_i k_i * P_i for a few points (host side, K9): points [n,18] Jacobian, scalars [n,4] canonical"""
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 18)
        sc = np.ascontiguousarray(scalars_canon, dtype=np.uint64).reshape(-1, 4)
        assert len(pts) == len(sc)
        out = np.zeros(18, dtype=np.uint64)
        self._check(self.lib.zk_g1_lincomb(self.h, _h(pts), _h(sc), len(pts), _h(out)))
        return out

    def g1_lincomb_batch(self, points: np.ndarray, scalars_canon: np.ndarray) -> np.ndarray:
        """points [count, n, 18], scalars [n, 4] shared by every row -> [count, 18]"""
        pts = np.ascontiguousarray(points, dtype=np.uint64)
        count, n = pts.shape[0], pts.shape[1]
        sc = np.ascontiguousarray(scalars_canon, dtype=np.uint64).reshape(-1, 4)
        assert len(sc) == n
        out = np.zeros((count, 18), dtype=np.uint64)
        if count:
            self._check(self.lib.zk_g1_lincomb_batch(self.h, _h(pts), _h(sc), n, count, _h(out)))
        return out

    # ---- party exchanges through the C ABI (RCCL communicator inside the ctx) ----
    def comm_unique_id(self) -> bytes:
        buf = (ctypes.c_uint8 * 128)()
        rc = self.lib.zk_comm_unique_id(buf)
        if rc != 0:
            raise ZkError(rc, "zk_comm_unique_id failed (RCCL not loadable?)")
        return bytes(buf)

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == 128
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.zk_comm_init(self.h, rank, world, buf))

    def comm_destroy(self):
        self._check(self.lib.zk_comm_destroy(self.h))

    def comm_abort(self):
        """this party cannot go on: its peers' pending and later collectives end with ZK_ERR_COMM instead of waiting for it (zk_comm_abort)"""
        self._check(self.lib.zk_comm_abort(self.h))

    @property
    def comm_rank(self) -> int:
        return self.lib.zk_comm_rank(self.h)

    @property
    def comm_size(self) -> int:
        return self.lib.zk_comm_size(self.h)

    def allgather(self, d_send, nbytes: int, d_recv=None):
        d_recv = d_recv or self.alloc(max(nbytes * self.comm_size, 1))
        self._check(self.lib.zk_allgather(self.h, _ptr(d_send), nbytes, _ptr(d_recv)))
        return d_recv

    def alltoall(self, d_send, nbytes_per_peer: int, d_recv=None):
        d_recv = d_recv or self.alloc(max(nbytes_per_peer * self.comm_size, 1))
        self._check(self.lib.zk_alltoall(self.h, _ptr(d_send), nbytes_per_peer, _ptr(d_recv)))
        return d_recv

    def gather(self, d_send, nbytes: int, root: int, d_recv=None):
        if self.comm_rank == root:
            d_recv = d_recv or self.alloc(max(nbytes * self.comm_size, 1))
        self._check(self.lib.zk_gather(self.h, _ptr(d_send), nbytes, root, _ptr(d_recv)))
        return d_recv

    def scatter(self, d_send, nbytes: int, root: int, d_recv=None):
        d_recv = d_recv or self.alloc(max(nbytes, 1))
        self._check(self.lib.zk_scatter(self.h, _ptr(d_send), nbytes, root, _ptr(d_recv)))
        return d_recv

    def d_msm(self, srs_list, scalars_list, lens, coeffs_canon: np.ndarray, lam_mont=None, offsets=None) -> np.ndarray:
        """zk_d_msm: batch of local MSMs + all-gather + this party's row of the public map -> [count, 18]"""
        count = len(lens)
        out = np.zeros((count, 18), dtype=np.uint64)
        if count == 0:
            return out
        h = (ctypes.c_void_p * count)(*[s.h for s in srs_list])
        sp = (ctypes.c_void_p * count)(*[_ptr(s) for s in scalars_list])
        nn = (ctypes.c_size_t * count)(*[int(x) for x in lens])
        off = (ctypes.c_size_t * count)(*[int(x) for x in (offsets or [0] * count)])
        co = np.ascontiguousarray(coeffs_canon, dtype=np.uint64).reshape(-1, 4)
        assert len(co) == self.comm_size
        lam = np.ascontiguousarray(lam_mont, dtype=np.uint64) if lam_mont is not None else None
        rc = self.lib.zk_d_msm(self.h, count, h, off, sp, nn, _h(lam) if lam is not None else 0, _h(co), _h(out))
        if rc == ZK_ERR_LENGTH:
            raise MsmLengthError(rc, (self.lib.zk_last_error(self.h) or b"").decode(), 0)
        self._check(rc)
        return out

    def msm_set_window(self, c: int):
        self._check(self.lib.zk_msm_set_window(self.h, c))

    def sumcheck_last_timing(self) -> np.ndarray:
        """[first stage ms, all launches ms] of the last sumcheck-family call (recorded while dbg_tune("sc_ts", 3))"""
        t = np.zeros(2, dtype=np.float32)
        self._check(self.lib.zk_sumcheck_last_timing(self.h, _h(t)))
        return t

    def msm_last_timing(self) -> np.ndarray:
        t = np.zeros(6, dtype=np.float32)
        self._check(self.lib.zk_msm_last_timing(self.h, _h(t)))
        return t

    # ---- test hooks (include/zkhip_test.h: not part of the ABI; resolved on first use) ----
    def dbg_tune(self, key: str, value: int):
        """process-wide experiment / diagnostics knob (zk_dbg_tune; csrc/zk_ctx.hpp `struct Tuning`)"""
        rc = test_hooks().zk_dbg_tune(key.encode(), int(value))
        if rc != 0:
            raise ZkError(rc, f"zk_dbg_tune: unknown key {key!r}")

    def dbg_fq(self, op: str, a, b, n, out=None):
        h = test_hooks()
        fn = {"add": h.zk_dbg_fq_add, "sub": h.zk_dbg_fq_sub, "mul": h.zk_dbg_fq_mul, "mul2add": h.zk_dbg_fq_mul2add}[op]
        out = out or self.alloc(max(48 * n, 1))
        self._check(fn(self.h, _ptr(a), _ptr(b), _ptr(out), n))
        return out

    def dbg_g2_op(self, mode: int, p, q, n) -> np.ndarray:
        out = np.zeros((n, 36), dtype=np.uint64)
        self._check(test_hooks().zk_dbg_g2_op(self.h, mode, _ptr(p), _ptr(q), _h(out), n))
        return out

    def dbg_g1_op(self, mode: int, p, q, n) -> np.ndarray:
        out = np.zeros((n, 18), dtype=np.uint64)
        self._check(test_hooks().zk_dbg_g1_op(self.h, mode, _ptr(p), _ptr(q), _h(out), n))
        return out
