/* zkhip_test.h -- TEST HOOKS of libzkhip.so, NOT ABI.
 *
 * The zk_dbg_* entry points exist for tests/ and tools/ only.  They are not part of the drop-in surface of include/zkhip.h: a
 * reference-side binding must not bind them (rust/zkhip_sys.rs is generated from zkhip.h alone and does not carry them), the
 * Python host resolves them only when a test or a tool asks (zkhip._lib.test_hooks()), and they may change or disappear
 * between versions. */
#ifndef ZKHIP_TEST_H
#define ZKHIP_TEST_H
#include "zkhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Process-wide experiment / diagnostics knobs (csrc/zk_ctx.hpp `struct Tuning` lists them; the same keys are read once
 * from ZKHIP_TUNE="key=value,..."): e.g. "sc_t1_device" = 1 makes the product sumcheck compute t1 = sum f_hi g_hi of
 * EVERY round on the device instead of deriving it from the previous round polynomial (the cross-check of
 * tests/test_gpu_bigsizes.py).  Returns ZK_ERR_INVALID for an unknown key. */
int zk_dbg_tune(const char *key, long value);
int zk_dbg_fq_mul(zk_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);
int zk_dbg_fq_add(zk_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);
int zk_dbg_fq_sub(zk_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);
/* a*b + b*b through the fused two-product multiplication (one Montgomery reduction) */
int zk_dbg_fq_mul2add(zk_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);
/* device XYZZ formulas on pairs of packed affine points; h_out[i] = 18 u64 normalised Jacobian.
 * mode 0: p+q (mixed add)  1: (p+q)+p (full add)  2: (p+q)+(p+q) (doubling path)  3: p-q */
int zk_dbg_g1_op(zk_ctx *ctx, int mode, const void *d_p96, const void *d_q96, void *h_out, size_t n);

/* the G2 formulas on pairs of affine points (192 B, reference form); h_out[i] = 36 u64 normalised Jacobian.
 * mode 0: p+q  1: (p+q)+p  2: (p+q)+(p+q) (full-addition doubling path)  3: p-q  4: 2(p+q) (doubling)  5: (p+q)-(p+q) */
int zk_dbg_g2_op(zk_ctx *ctx, int mode, const void *d_p192, const void *d_q192, void *h_out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_TEST_H */
