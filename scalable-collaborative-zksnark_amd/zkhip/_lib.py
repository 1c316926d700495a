"""
ctypes binding of libzkhip.so (include/zkhip.h).  The library is the product; there is no
Python/CPU fallback: if it is missing or has no usable GPU, loading/creating a context raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get("ZKHIP_LIB") or os.path.join(_ROOT, "libzkhip.so")  # ZKHIP_LIB: an explicitly chosen build (A/B runs)
CSRC = os.path.join(_ROOT, "csrc")

# every symbol include/zkhip.h declares: (name, restype, argtypes)
_vp, _sz, _i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
_pp = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = [
    ("zk_ctx_create", _i, [_i, _pp]),
    ("zk_ctx_destroy", None, [_vp]),
    ("zk_last_error", ctypes.c_char_p, [_vp]),
    ("zk_ctx_set_stream", _i, [_vp, _vp]),
    ("zk_ctx_sync", _i, [_vp]),
    ("zk_version", ctypes.c_char_p, []),
    ("zk_device_count", _i, []),
    ("zk_mem_info", _i, [_vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    ("zk_malloc", _i, [_vp, _sz, _pp]),
    ("zk_free", _i, [_vp, _vp]),
    ("zk_trim", _i, [_vp, ctypes.POINTER(ctypes.c_size_t)]),
    ("zk_memcpy_h2d", _i, [_vp, _vp, _vp, _sz]),
    ("zk_memcpy_d2h", _i, [_vp, _vp, _vp, _sz]),
    ("zk_memcpy_d2d", _i, [_vp, _vp, _vp, _sz]),
    ("zk_fr_add", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_fr_sub", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_fr_mul", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_fr_axpb", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    ("zk_fr_apply_matrix", _i, [_vp, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _sz]),
    ("zk_fr_ntt_map", _i, [_vp, _sz, _vp, _sz, _vp, _vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _sz]),
    ("zk_fr_deinterleave", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_fr_batch_div", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_sumcheck", _i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    ("zk_sumcheck_product", _i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    ("zk_fold", _i, [_vp, _vp, _sz, _vp, _sz, _vp]),
    ("zk_open_rounds", _i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    ("zk_sumcheck_batch", _i, [_vp, _sz, _vp]),
    ("zk_product_tree", _i, [_vp, _vp, _sz, _vp]),
    ("zk_srs_register", _i, [_vp, _vp, _sz, _sz, _pp]),
    ("zk_srs_wrap_device", _i, [_vp, _vp, _sz, _pp]),
    ("zk_srs_generate", _i, [_vp, _vp, _vp, _sz, _pp]),
    ("zk_srs_powers", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("zk_srs_to_packed", _i, [_vp, _vp, _vp, _sz, _pp]),
    ("zk_g1_apply_matrix", _i, [_vp, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _sz]),
    ("zk_srs_precompute", _i, [_vp, _vp, _i]),
    ("zk_srs_precompute_layout", _i, [_vp, _vp, _i, _i]),
    ("zk_srs_table_window", _i, [_vp]),
    ("zk_srs_table_record", _i, [_vp]),
    ("zk_srs_free", _i, [_vp, _vp]),
    ("zk_srs_len", _sz, [_vp]),
    ("zk_srs_device_ptr", _vp, [_vp]),
    ("zk_srs_download", _i, [_vp, _vp, _vp]),
    ("zk_msm_g1", _i, [_vp, _vp, _sz, _vp, _sz, _vp]),
    ("zk_msm_g1_batch", _i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("zk_msm_g1_batch_async", _i, [_vp, _sz, _vp, _vp, _vp, _vp, _pp]),
    ("zk_msm_wait", _i, [_vp, _vp, _vp]),
    ("zk_srs_register_g2", _i, [_vp, _vp, _sz, _sz, _pp]),
    ("zk_msm_g2", _i, [_vp, _vp, _sz, _vp, _sz, _vp]),
    ("zk_msm_g2_batch", _i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("zk_msm_g1_host", _i, [_vp, _vp, _sz, _sz, _vp, _sz, _vp, ctypes.POINTER(ctypes.c_size_t)]),
    ("zk_g1_lincomb", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("zk_g1_lincomb_batch", _i, [_vp, _vp, _vp, _sz, _sz, _vp]),
    ("zk_msm_window", _i, [_sz]),
    ("zk_msm_set_window", _i, [_vp, _i]),
    ("zk_msm_last_timing", _i, [_vp, _vp]),
    ("zk_sumcheck_last_timing", _i, [_vp, _vp]),
    ("zk_comm_unique_id", _i, [_vp]),
    ("zk_comm_init", _i, [_vp, _i, _i, _vp]),
    ("zk_comm_init_all", _i, [_vp, _i]),
    ("zk_comm_destroy", _i, [_vp]),
    ("zk_comm_abort", _i, [_vp]),
    ("zk_comm_rank", _i, [_vp]),
    ("zk_comm_size", _i, [_vp]),
    ("zk_allgather", _i, [_vp, _vp, _sz, _vp]),
    ("zk_alltoall", _i, [_vp, _vp, _sz, _vp]),
    ("zk_gather", _i, [_vp, _vp, _sz, _i, _vp]),
    ("zk_scatter", _i, [_vp, _vp, _sz, _i, _vp]),
    ("zk_d_msm", _i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("zk_arena_plan_export", _i, [_vp, _vp]),
    ("zk_arena_plan_import", _i, [_vp, _vp]),
]

# include/zkhip_test.h: the zk_dbg_* test hooks -- not part of the ABI, resolved only when a test / tool asks (test_hooks())
TEST_SYMBOLS = [
    ("zk_dbg_tune", _i, [ctypes.c_char_p, ctypes.c_long]),
    ("zk_dbg_fq_mul", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_dbg_fq_add", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_dbg_fq_sub", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_dbg_fq_mul2add", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("zk_dbg_g1_op", _i, [_vp, _i, _vp, _vp, _vp, _sz]),
    ("zk_dbg_g2_op", _i, [_vp, _i, _vp, _vp, _vp, _sz]),
]

class ScItem(ctypes.Structure):
    """include/zkhip.h `zk_sc_item`"""

    _fields_ = [("mode", _i), ("d_f", _vp), ("d_g", _vp), ("len", _sz), ("h_chal", _vp), ("n_points", _sz), ("h_sums", _vp), ("h_last_f", _vp),
                ("h_last_g", _vp), ("d_out", _vp)]


ZK_OK, ZK_ERR_INVALID, ZK_ERR_LENGTH, ZK_ERR_HIP, ZK_ERR_NO_DEVICE, ZK_ERR_DIV_ZERO, ZK_ERR_OOM, ZK_ERR_COMM = 0, -1, -2, -3, -4, -5, -6, -7


def build(force: bool = False) -> str:
    """compile libzkhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)"""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    """load libzkhip.so; raises if it is not built (no fallback path exists)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (or `make -C {CSRC}`). "
                "zkhip has no CPU fallback."
            )
        l = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(l, name)  # AttributeError here = header/library mismatch
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


_hooks = False


def test_hooks() -> ctypes.CDLL:
    """the library with the zk_dbg_* test hooks of include/zkhip_test.h resolved as well (tests/, tools/, bench.py's timing knobs)"""
    global _hooks
    l = lib()
    if not _hooks:
        for name, res, args in TEST_SYMBOLS:
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _hooks = True
    return l
