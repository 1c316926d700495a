/*
 * zk_oracle.c -- plain-C CPU restatement of the reference's dist-primitive hot
 * path (single-threaded, exactly like the reference: arkworks is pulled without
 * its `parallel` feature, Cargo.lock:120-134).
 *
 * TEST INFRASTRUCTURE ONLY.  Linked/loaded only by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, as the checker / the timed CPU "port".  The
 * product library (libzkhip.so) never links or calls this file.
 *
 * PARITY STATUS: parity unpinned for MSM / sumcheck / fold (the reference is
 * Rust on un-vendored arkworks 0.4.x and holds no golden vectors for them);
 * pinned for the product tree by the reference KAT dacc_product.rs:450-466.
 * This file is itself checked against oracle/pyoracle.py (pure big-int) by
 * tests/test_oracle_c.py.
 *
 * Memory layouts are the reference's (ark-ff 0.4.2 MontBackend): Fr = 4xu64,
 * Fq = 6xu64 little-endian limbs in Montgomery form; affine G1 point = x||y
 * (96 B) with x=y=0 meaning infinity (not on y^2=x^3+4).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ */
/* generic Montgomery arithmetic (CIOS), N limbs                       */
/* ------------------------------------------------------------------ */
#define NR 4
#define NQ 6

static const u64 FR_P[NR] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const u64 FR_INV = 0xfffffffeffffffffULL; /* -r^{-1} mod 2^64 */
static const u64 FR_ONE[NR] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}; /* R mod r */
static const u64 FR_R2[NR] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}; /* R^2 mod r */

static const u64 FQ_P[NQ] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                             0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 FQ_INV = 0x89f3fffcfffcfffdULL;
static const u64 FQ_ONE[NQ] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                               0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const u64 FQ_R2[NQ] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                              0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};

static inline __attribute__((always_inline)) int ge_n(const u64 *a, const u64 *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline __attribute__((always_inline)) u64 sub_n(u64 *r, const u64 *a, const u64 *b, int n) {
    u64 borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)t;
        borrow = (u64)(t >> 64) & 1;
    }
    return borrow;
}
static inline __attribute__((always_inline)) u64 add_n(u64 *r, const u64 *a, const u64 *b, int n) {
    u64 carry = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] + b[i] + carry;
        r[i] = (u64)t;
        carry = (u64)(t >> 64);
    }
    return carry;
}
static inline __attribute__((always_inline)) void mod_add(u64 *r, const u64 *a, const u64 *b, const u64 *p, int n) {
    u64 t[NQ], d[NQ];
    u64 c = add_n(t, a, b, n);
    u64 bw = sub_n(d, t, p, n);
    u64 keep = (u64)0 - (u64)((bw != 0) & (c == 0)); /* all ones: keep t (t < p) */
#pragma GCC unroll 6
    for (int i = 0; i < n; i++) r[i] = (t[i] & keep) | (d[i] & ~keep);
}
static inline __attribute__((always_inline)) void mod_sub(u64 *r, const u64 *a, const u64 *b, const u64 *p, int n) {
    u64 t[NQ], pm[NQ];
    u64 bw = sub_n(t, a, b, n);
    u64 mask = (u64)0 - bw;
#pragma GCC unroll 6
    for (int i = 0; i < n; i++) pm[i] = p[i] & mask;
    add_n(r, t, pm, n);
}
static inline int is_zero_n(const u64 *a, int n) {
    u64 o = 0;
    for (int i = 0; i < n; i++) o |= a[i];
    return o == 0;
}
static inline void mod_neg(u64 *r, const u64 *a, const u64 *p, int n) {
    if (is_zero_n(a, n)) memset(r, 0, 8 * n);
    else sub_n(r, p, a, n);
}
/* CIOS Montgomery multiplication, fully unrolled with scalar accumulators (the generic
 * array/loop form is ~5x slower under gcc for 6 limbs, which would sandbag the CPU
 * baseline). */
#define MM_STEP(k, kp) x = (u128)a[k] * bi + t##k + c; t##k = (u64)x; c = (u64)(x >> 64);
#define MR_STEP(k, km) x = (u128)m * p[k] + t##k + c; t##km = (u64)x; c = (u64)(x >> 64);
static inline __attribute__((always_inline)) void mont_mul6(u64 *r, const u64 *a, const u64 *b, const u64 *p, u64 inv) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
#pragma GCC unroll 6
    for (int i = 0; i < 6; i++) {
        u64 bi = b[i], c = 0;
        u128 x;
        MM_STEP(0, 1) MM_STEP(1, 2) MM_STEP(2, 3) MM_STEP(3, 4) MM_STEP(4, 5) MM_STEP(5, 6)
        x = (u128)t6 + c; t6 = (u64)x; t7 = (u64)(x >> 64);
        u64 m = t0 * inv;
        x = (u128)m * p[0] + t0; c = (u64)(x >> 64);
        MR_STEP(1, 0) MR_STEP(2, 1) MR_STEP(3, 2) MR_STEP(4, 3) MR_STEP(5, 4)
        x = (u128)t6 + c; t5 = (u64)x; t6 = t7 + (u64)(x >> 64);
    }
    u64 t[6] = {t0, t1, t2, t3, t4, t5};
    if (t6 || ge_n(t, p, 6)) sub_n(t, t, p, 6);
    memcpy(r, t, 48);
}
static inline __attribute__((always_inline)) void mont_mul4(u64 *r, const u64 *a, const u64 *b, const u64 *p, u64 inv) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
#pragma GCC unroll 4
    for (int i = 0; i < 4; i++) {
        u64 bi = b[i], c = 0;
        u128 x;
        MM_STEP(0, 1) MM_STEP(1, 2) MM_STEP(2, 3) MM_STEP(3, 4)
        x = (u128)t4 + c; t4 = (u64)x; t5 = (u64)(x >> 64);
        u64 m = t0 * inv;
        x = (u128)m * p[0] + t0; c = (u64)(x >> 64);
        MR_STEP(1, 0) MR_STEP(2, 1) MR_STEP(3, 2)
        x = (u128)t4 + c; t3 = (u64)x; t4 = t5 + (u64)(x >> 64);
    }
    u64 t[4] = {t0, t1, t2, t3};
    if (t4 || ge_n(t, p, 4)) sub_n(t, t, p, 4);
    memcpy(r, t, 32);
}

/* Fr / Fq front-ends */
#define FR_MUL(r, a, b) mont_mul4(r, a, b, FR_P, FR_INV)
#define FR_ADD(r, a, b) mod_add(r, a, b, FR_P, NR)
#define FR_SUB(r, a, b) mod_sub(r, a, b, FR_P, NR)
#define FQ_MUL(r, a, b) mont_mul6(r, a, b, FQ_P, FQ_INV)
#define FQ_ADD(r, a, b) mod_add(r, a, b, FQ_P, NQ)
#define FQ_SUB(r, a, b) mod_sub(r, a, b, FQ_P, NQ)

static void fr_pow(u64 *r, const u64 *a, const u64 *e, int elimbs) {
    u64 acc[NR], base[NR];
    memcpy(acc, FR_ONE, 32);
    memcpy(base, a, 32);
    for (int i = elimbs * 64 - 1; i >= 0; i--) {
        FR_MUL(acc, acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) FR_MUL(acc, acc, base);
    }
    memcpy(r, acc, 32);
}
static void fr_inv(u64 *r, const u64 *a) { /* a^(r-2); 0 -> 0 */
    u64 e[NR];
    u64 two[NR] = {2, 0, 0, 0};
    sub_n(e, FR_P, two, NR);
    fr_pow(r, a, e, NR);
}
static void fq_inv(u64 *r, const u64 *a) {
    u64 e[NQ], two[NQ] = {2, 0, 0, 0, 0, 0};
    sub_n(e, FQ_P, two, NQ);
    u64 acc[NQ], base[NQ];
    memcpy(acc, FQ_ONE, 48);
    memcpy(base, a, 48);
    for (int i = NQ * 64 - 1; i >= 0; i--) {
        FQ_MUL(acc, acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) FQ_MUL(acc, acc, base);
    }
    memcpy(r, acc, 48);
}

/* ------------------------------------------------------------------ */
/* exported Fr vector helpers                                          */
/* ------------------------------------------------------------------ */
void ora_fr_to_mont(const u64 *canon, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FR_MUL(out + 4 * i, canon + 4 * i, FR_R2);
}
void ora_fr_from_mont(const u64 *mont, u64 *out, size_t n) {
    u64 one[NR] = {1, 0, 0, 0};
    for (size_t i = 0; i < n; i++) FR_MUL(out + 4 * i, mont + 4 * i, one);
}
void ora_fq_to_mont(const u64 *canon, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FQ_MUL(out + 6 * i, canon + 6 * i, FQ_R2);
}
void ora_fq_from_mont(const u64 *mont, u64 *out, size_t n) {
    u64 one[NQ] = {1, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) FQ_MUL(out + 6 * i, mont + 6 * i, one);
}
void ora_fr_mul(const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FR_MUL(out + 4 * i, a + 4 * i, b + 4 * i);
}
void ora_fr_add(const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FR_ADD(out + 4 * i, a + 4 * i, b + 4 * i);
}
void ora_fr_sub(const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FR_SUB(out + 4 * i, a + 4 * i, b + 4 * i);
}
void ora_fq_mul(const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FQ_MUL(out + 6 * i, a + 6 * i, b + 6 * i);
}
void ora_fq_add(const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FQ_ADD(out + 6 * i, a + 6 * i, b + 6 * i);
}
void ora_fq_sub(const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) FQ_SUB(out + 6 * i, a + 6 * i, b + 6 * i);
}
/* out = a / b element-wise, as `num / den` in dhyperplonk.rs:339 (inverse per element).
 * returns -1 if some b is zero (the reference would panic on inverse().unwrap()) */
int ora_fr_div(const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (is_zero_n(b + 4 * i, NR)) return -1;
        u64 bi[NR];
        fr_inv(bi, b + 4 * i);
        FR_MUL(out + 4 * i, a + 4 * i, bi);
    }
    return 0;
}

/* SplitMix64 -> uniform Fr (same stream as pyoracle.SplitMix64.fr); output limbs are
 * written as-is (they are then *interpreted* as Montgomery limbs by callers) */
static u64 sm_next(u64 *s) {
    *s += 0x9E3779B97F4A7C15ULL;
    u64 z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void ora_rand_fr(u64 seed, u64 *out, size_t n) {
    u64 s = seed;
    for (size_t i = 0; i < n; i++) {
        u64 v[NR];
        do {
            for (int k = 0; k < NR; k++) v[k] = sm_next(&s);
            v[3] &= 0x7fffffffffffffffULL;
        } while (ge_n(v, FR_P, NR));
        memcpy(out + 4 * i, v, 32);
    }
}

/* ------------------------------------------------------------------ */
/* multilinear loops                                                   */
/* ------------------------------------------------------------------ */
/* new[j] = lo[j]*(1-r) + hi[j]*r  -- the formula as the reference writes it
 * (dsumcheck.rs:14-19): two multiplications and one addition per element */
static void fold_ref(const u64 *tab, size_t m, const u64 *r, u64 *out) {
    u64 omr[NR];
    FR_SUB(omr, FR_ONE, r);
    size_t h = m / 2;
    for (size_t j = 0; j < h; j++) {
        u64 a[NR], b[NR];
        FR_MUL(a, tab + 4 * j, omr);
        FR_MUL(b, tab + 4 * (j + h), r);
        FR_ADD(out + 4 * j, a, b);
    }
}
void ora_fold(const u64 *tab, size_t m, const u64 *r, u64 *out) { fold_ref(tab, m, r, out); }

/* dsumcheck.rs:6-26.  out: (n+1) pairs = 2*(n+1) Fr */
void ora_sumcheck(const u64 *tab, size_t len, const u64 *chal, u64 *out) {
    int n = __builtin_ctzll(len);
    u64 *cur = malloc(32 * len), *nxt = malloc(16 * len + 32);
    memcpy(cur, tab, 32 * len);
    size_t m = len;
    for (int i = 0; i < n; i++) {
        size_t h = m / 2;
        u64 s0[NR] = {0}, s1[NR] = {0};
        for (size_t j = 0; j < h; j++) FR_ADD(s0, s0, cur + 4 * j);
        for (size_t j = h; j < m; j++) FR_ADD(s1, s1, cur + 4 * j);
        memcpy(out + 8 * i, s0, 32);
        memcpy(out + 8 * i + 4, s1, 32);
        fold_ref(cur, m, chal + 4 * i, nxt);
        u64 *t = cur; cur = nxt; nxt = t;
        m = h;
    }
    memset(out + 8 * n, 0, 32);
    memcpy(out + 8 * n + 4, cur, 32);
    free(cur); free(nxt);
}

/* the Phase-1 loop shared by sumcheck_product / c_ / d_ variants
 * (dsumcheck.rs:37-85 = :167-219 = :377-429).  out: n triples; last_f/last_g = the
 * single remaining element of each table. */
void ora_sumcheck_product_rounds(const u64 *f, const u64 *g, size_t len, const u64 *chal, u64 *out,
                                 u64 *last_f, u64 *last_g) {
    int n = __builtin_ctzll(len);
    u64 *cf = malloc(32 * len), *cg = malloc(32 * len), *nf = malloc(16 * len + 32), *ng = malloc(16 * len + 32);
    memcpy(cf, f, 32 * len);
    memcpy(cg, g, 32 * len);
    size_t m = len;
    u64 two[NR];
    FR_ADD(two, FR_ONE, FR_ONE);
    for (int i = 0; i < n; i++) {
        size_t h = m / 2;
        u64 t0[NR] = {0}, t1[NR] = {0}, t2[NR] = {0}, x[NR], y[NR];
        for (size_t j = 0; j < h; j++) {
            FR_MUL(x, cf + 4 * j, cg + 4 * j);
            FR_ADD(t0, t0, x);
        }
        for (size_t j = h; j < m; j++) {
            FR_MUL(x, cf + 4 * j, cg + 4 * j);
            FR_ADD(t1, t1, x);
        }
        for (size_t j = 0; j < h; j++) { /* -x + y*2, as written at :55-58 */
            u64 pf[NR], pg[NR];
            FR_MUL(x, cf + 4 * (j + h), two);
            FR_SUB(pf, x, cf + 4 * j);
            FR_MUL(y, cg + 4 * (j + h), two);
            FR_SUB(pg, y, cg + 4 * j);
            FR_MUL(x, pf, pg);
            FR_ADD(t2, t2, x);
        }
        memcpy(out + 12 * i, t0, 32);
        memcpy(out + 12 * i + 4, t1, 32);
        memcpy(out + 12 * i + 8, t2, 32);
        fold_ref(cf, m, chal + 4 * i, nf);
        fold_ref(cg, m, chal + 4 * i, ng);
        u64 *t = cf; cf = nf; nf = t;
        t = cg; cg = ng; ng = t;
        m = h;
    }
    memcpy(last_f, cf, 32);
    memcpy(last_g, cg, 32);
    free(cf); free(cg); free(nf); free(ng);
}

/* dsumcheck.rs:28-90: n+1 triples, last = (0, f*g, 0) */
void ora_sumcheck_product(const u64 *f, const u64 *g, size_t len, const u64 *chal, u64 *out) {
    int n = __builtin_ctzll(len);
    u64 lf[NR], lg[NR];
    ora_sumcheck_product_rounds(f, g, len, chal, out, lf, lg);
    memset(out + 12 * n, 0, 96);
    FR_MUL(out + 12 * n + 4, lf, lg);
}

/* phase 1 of open / c_open (dpoly_comm.rs:309-323 = :418-432): q_i = hi - lo for every
 * round, concatenated (len-1 elements: len/2, len/4, ..., 1), plus the final value */
void ora_open_quotients(const u64 *tab, size_t len, const u64 *point, u64 *q_out, u64 *value) {
    int n = __builtin_ctzll(len);
    u64 *cur = malloc(32 * len), *nxt = malloc(16 * len + 32);
    memcpy(cur, tab, 32 * len);
    size_t m = len, off = 0;
    for (int i = 0; i < n; i++) {
        size_t h = m / 2;
        for (size_t j = 0; j < h; j++) FR_SUB(q_out + 4 * (off + j), cur + 4 * (j + h), cur + 4 * j);
        off += h;
        fold_ref(cur, m, point + 4 * i, nxt);
        u64 *t = cur; cur = nxt; nxt = t;
        m = h;
    }
    memcpy(value, cur, 32);
    free(cur); free(nxt);
}

/* dacc_product.rs:31-38: tree (2N elements) */
void ora_product_tree(const u64 *x, size_t N, u64 *tree) {
    memcpy(tree, x, 32 * N);
    memcpy(tree + 4 * N, x, 32 * N);
    for (size_t i = N; i < 2 * N - 1; i++) {
        int top = 63 - __builtin_clzll(i);
        size_t a = (i & ~((size_t)1 << top)) << 1; /* sub_index :18-23 */
        FR_MUL(tree + 4 * i, tree + 4 * a, tree + 4 * (a + 1));
    }
    memset(tree + 4 * (2 * N - 1), 0, 32);
}

/* ------------------------------------------------------------------ */
/* G1: Jacobian arithmetic as ark-ec short_weierstrass::Projective     */
/* ------------------------------------------------------------------ */
typedef struct { u64 x[NQ], y[NQ], z[NQ]; } jac_t;
typedef struct { u64 x[NQ], y[NQ]; } aff_t; /* x=y=0 : infinity */

static inline int aff_is_inf(const aff_t *p) { return is_zero_n(p->x, NQ) && is_zero_n(p->y, NQ); }
static inline void jac_set_inf(jac_t *p) {
    memcpy(p->x, FQ_ONE, 48);
    memcpy(p->y, FQ_ONE, 48);
    memset(p->z, 0, 48);
}
static inline int jac_is_inf(const jac_t *p) { return is_zero_n(p->z, NQ); }

static void jac_double(jac_t *r, const jac_t *p) { /* dbl-2009-l, a = 0 */
    if (jac_is_inf(p)) { *r = *p; return; }
    u64 A[NQ], B[NQ], C[NQ], D[NQ], E[NQ], F[NQ], t[NQ];
    FQ_MUL(A, p->x, p->x);
    FQ_MUL(B, p->y, p->y);
    FQ_MUL(C, B, B);
    FQ_ADD(t, p->x, B);
    FQ_MUL(t, t, t);
    FQ_SUB(t, t, A);
    FQ_SUB(t, t, C);
    FQ_ADD(D, t, t);
    FQ_ADD(E, A, A);
    FQ_ADD(E, E, A);
    FQ_MUL(F, E, E);
    u64 z3[NQ];
    FQ_MUL(z3, p->y, p->z);
    FQ_ADD(z3, z3, z3);
    FQ_SUB(r->x, F, D);
    FQ_SUB(r->x, r->x, D);
    FQ_SUB(t, D, r->x);
    FQ_MUL(t, E, t);
    FQ_ADD(C, C, C);
    FQ_ADD(C, C, C);
    FQ_ADD(C, C, C);
    FQ_SUB(r->y, t, C);
    memcpy(r->z, z3, 48);
}
static void jac_add_mixed(jac_t *r, const jac_t *p, const aff_t *q) { /* madd-2007-bl */
    if (aff_is_inf(q)) { *r = *p; return; }
    if (jac_is_inf(p)) {
        memcpy(r->x, q->x, 48);
        memcpy(r->y, q->y, 48);
        memcpy(r->z, FQ_ONE, 48);
        return;
    }
    u64 Z1Z1[NQ], U2[NQ], S2[NQ], H[NQ], HH[NQ], I[NQ], J[NQ], rr[NQ], V[NQ], t[NQ];
    FQ_MUL(Z1Z1, p->z, p->z);
    FQ_MUL(U2, q->x, Z1Z1);
    FQ_MUL(S2, q->y, p->z);
    FQ_MUL(S2, S2, Z1Z1);
    if (memcmp(U2, p->x, 48) == 0) {
        if (memcmp(S2, p->y, 48) == 0) { jac_double(r, p); return; }
        jac_set_inf(r);
        return;
    }
    FQ_SUB(H, U2, p->x);
    FQ_MUL(HH, H, H);
    FQ_ADD(I, HH, HH);
    FQ_ADD(I, I, I);
    FQ_MUL(J, H, I);
    FQ_SUB(rr, S2, p->y);
    FQ_ADD(rr, rr, rr);
    FQ_MUL(V, p->x, I);
    u64 x3[NQ], y3[NQ], z3[NQ];
    FQ_MUL(x3, rr, rr);
    FQ_SUB(x3, x3, J);
    FQ_SUB(x3, x3, V);
    FQ_SUB(x3, x3, V);
    FQ_SUB(t, V, x3);
    FQ_MUL(y3, rr, t);
    FQ_MUL(t, p->y, J);
    FQ_ADD(t, t, t);
    FQ_SUB(y3, y3, t);
    FQ_ADD(z3, p->z, H);
    FQ_MUL(z3, z3, z3);
    FQ_SUB(z3, z3, Z1Z1);
    FQ_SUB(z3, z3, HH);
    memcpy(r->x, x3, 48);
    memcpy(r->y, y3, 48);
    memcpy(r->z, z3, 48);
}
static void jac_add(jac_t *r, const jac_t *p, const jac_t *q) { /* add-2007-bl */
    if (jac_is_inf(p)) { *r = *q; return; }
    if (jac_is_inf(q)) { *r = *p; return; }
    u64 Z1Z1[NQ], Z2Z2[NQ], U1[NQ], U2[NQ], S1[NQ], S2[NQ], H[NQ], I[NQ], J[NQ], rr[NQ], V[NQ], t[NQ];
    FQ_MUL(Z1Z1, p->z, p->z);
    FQ_MUL(Z2Z2, q->z, q->z);
    FQ_MUL(U1, p->x, Z2Z2);
    FQ_MUL(U2, q->x, Z1Z1);
    FQ_MUL(S1, p->y, q->z);
    FQ_MUL(S1, S1, Z2Z2);
    FQ_MUL(S2, q->y, p->z);
    FQ_MUL(S2, S2, Z1Z1);
    if (memcmp(U1, U2, 48) == 0) {
        if (memcmp(S1, S2, 48) == 0) { jac_double(r, p); return; }
        jac_set_inf(r);
        return;
    }
    FQ_SUB(H, U2, U1);
    FQ_ADD(I, H, H);
    FQ_MUL(I, I, I);
    FQ_MUL(J, H, I);
    FQ_SUB(rr, S2, S1);
    FQ_ADD(rr, rr, rr);
    FQ_MUL(V, U1, I);
    u64 x3[NQ], y3[NQ], z3[NQ];
    FQ_MUL(x3, rr, rr);
    FQ_SUB(x3, x3, J);
    FQ_SUB(x3, x3, V);
    FQ_SUB(x3, x3, V);
    FQ_SUB(t, V, x3);
    FQ_MUL(y3, rr, t);
    FQ_MUL(t, S1, J);
    FQ_ADD(t, t, t);
    FQ_SUB(y3, y3, t);
    FQ_ADD(z3, p->z, q->z);
    FQ_MUL(z3, z3, z3);
    FQ_SUB(z3, z3, Z1Z1);
    FQ_SUB(z3, z3, Z2Z2);
    FQ_MUL(z3, z3, H);
    memcpy(r->x, x3, 48);
    memcpy(r->y, y3, 48);
    memcpy(r->z, z3, 48);
}
static void jac_to_affine(aff_t *r, const jac_t *p) {
    if (jac_is_inf(p)) { memset(r, 0, sizeof(*r)); return; }
    u64 zi[NQ], zi2[NQ];
    fq_inv(zi, p->z);
    FQ_MUL(zi2, zi, zi);
    FQ_MUL(r->x, p->x, zi2);
    FQ_MUL(zi2, zi2, zi);
    FQ_MUL(r->y, p->y, zi2);
}

/* ------------------------------------------------------------------ */
/* VariableBaseMSM::msm as ark-ec 0.4.2 does it for G1 (restated from the
 * published algorithm; the crate is not in /root/reference): scalars out of
 * Montgomery form, signed digits (make_digits), window c = 3 if n < 32 else
 * ceil_log2(n)*69/100 + 2, 1<<c buckets, running-sum reduction, windows
 * combined high->low with c doublings.  Single thread.                 */
/* ------------------------------------------------------------------ */
static int ceil_log2(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}
int ora_msm_window(size_t n) { return n < 32 ? 3 : (ceil_log2(n) * 69 / 100) + 2; }

int ora_msm_g1(const u64 *bases_affine /* 12 u64 per point */, const u64 *scalars_mont, size_t n, u64 *out_affine) {
    aff_t res;
    if (n == 0) { memset(out_affine, 0, 96); return 0; }
    const aff_t *bases = (const aff_t *)bases_affine;
    int c = ora_msm_window(n);
    int num_bits = 255;
    int digits_count = (num_bits + c - 1) / c;
    int64_t *digits = malloc(sizeof(int64_t) * n * digits_count);
    u64 one[NR] = {1, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        u64 s[NR];
        FR_MUL(s, scalars_mont + 4 * i, one); /* into_bigint */
        u64 radix = 1ULL << c, mask = radix - 1, carry = 0;
        int64_t *d = digits + i * digits_count;
        for (int k = 0; k < digits_count; k++) {
            int bit_offset = k * c, idx = bit_offset / 64, bit = bit_offset % 64;
            u64 buf;
            if (bit < 64 - c || idx == NR - 1) buf = s[idx] >> bit;
            else buf = (s[idx] >> bit) | (s[idx + 1] << (64 - bit));
            u64 coef = carry + (buf & mask);
            carry = (coef + radix / 2) >> c;
            d[k] = (int64_t)coef - (int64_t)(carry << c);
        }
        d[digits_count - 1] += (int64_t)(carry << c);
    }
    jac_t *window_sums = malloc(sizeof(jac_t) * digits_count);
    jac_t *buckets = malloc(sizeof(jac_t) * ((size_t)1 << c));
    for (int w = 0; w < digits_count; w++) {
        for (size_t b = 0; b < ((size_t)1 << c); b++) jac_set_inf(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int64_t sc = digits[i * digits_count + w];
            if (sc > 0) jac_add_mixed(&buckets[sc - 1], &buckets[sc - 1], &bases[i]);
            else if (sc < 0) {
                aff_t neg = bases[i];
                if (!aff_is_inf(&neg)) mod_neg(neg.y, neg.y, FQ_P, NQ);
                jac_add_mixed(&buckets[-sc - 1], &buckets[-sc - 1], &neg);
            }
        }
        jac_t running, r;
        jac_set_inf(&running);
        jac_set_inf(&r);
        for (size_t b = ((size_t)1 << c); b-- > 0;) {
            jac_add(&running, &running, &buckets[b]);
            jac_add(&r, &r, &running);
        }
        window_sums[w] = r;
    }
    jac_t total;
    jac_set_inf(&total);
    for (int w = digits_count - 1; w >= 1; w--) {
        jac_add(&total, &total, &window_sums[w]);
        for (int k = 0; k < c; k++) jac_double(&total, &total);
    }
    jac_add(&total, &total, &window_sums[0]);
    jac_to_affine(&res, &total);
    memcpy(out_affine, &res, 96);
    free(digits); free(window_sums); free(buckets);
    return 0;
}

/* affine helpers for tests */
void ora_g1_add_affine(const u64 *p, const u64 *q, u64 *out) {
    jac_t a, r;
    aff_t o;
    const aff_t *pa = (const aff_t *)p;
    if (aff_is_inf(pa)) jac_set_inf(&a);
    else { memcpy(a.x, pa->x, 48); memcpy(a.y, pa->y, 48); memcpy(a.z, FQ_ONE, 48); }
    jac_add_mixed(&r, &a, (const aff_t *)q);
    jac_to_affine(&o, &r);
    memcpy(out, &o, 96);
}
/* k*P for canonical (non-Montgomery) 4-limb scalar k */
void ora_g1_mul_affine(const u64 *p, const u64 *k_canon, u64 *out) {
    jac_t acc;
    aff_t o;
    jac_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        jac_double(&acc, &acc);
        if ((k_canon[i / 64] >> (i % 64)) & 1) jac_add_mixed(&acc, &acc, (const aff_t *)p);
    }
    jac_to_affine(&o, &acc);
    memcpy(out, &o, 96);
}
/* Jacobian (18 u64, Montgomery) -> affine (12 u64) */
void ora_g1_jac_to_affine(const u64 *jac, u64 *out) {
    aff_t o;
    jac_to_affine(&o, (const jac_t *)jac);
    memcpy(out, &o, 96);
}

/* synthetic SRS: P_i = start + i*step (affine inputs), batch-normalised.  Mirrors
 * pyoracle.g1_bases once start = k0*G and step = k1*G are supplied. */
void ora_g1_arith_seq(const u64 *start_aff, const u64 *step_aff, size_t n, u64 *out_affine) {
    jac_t *pts = malloc(sizeof(jac_t) * n);
    u64 *pref = malloc(48 * n);
    jac_t cur;
    const aff_t *st = (const aff_t *)start_aff;
    memcpy(cur.x, st->x, 48); memcpy(cur.y, st->y, 48); memcpy(cur.z, FQ_ONE, 48);
    if (aff_is_inf(st)) jac_set_inf(&cur);
    u64 acc[NQ];
    memcpy(acc, FQ_ONE, 48);
    for (size_t i = 0; i < n; i++) {
        pts[i] = cur;
        memcpy(pref + 6 * i, acc, 48);
        if (!jac_is_inf(&cur)) FQ_MUL(acc, acc, cur.z);
        jac_add_mixed(&cur, &cur, (const aff_t *)step_aff);
    }
    u64 inv[NQ];
    fq_inv(inv, acc);
    aff_t *out = (aff_t *)out_affine;
    for (size_t i = n; i-- > 0;) {
        if (jac_is_inf(&pts[i])) { memset(&out[i], 0, 96); continue; }
        u64 zi[NQ], zi2[NQ];
        FQ_MUL(zi, inv, pref + 6 * i);
        FQ_MUL(inv, inv, pts[i].z);
        FQ_MUL(zi2, zi, zi);
        FQ_MUL(out[i].x, pts[i].x, zi2);
        FQ_MUL(zi2, zi2, zi);
        FQ_MUL(out[i].y, pts[i].y, zi2);
    }
    free(pts); free(pref);
}
