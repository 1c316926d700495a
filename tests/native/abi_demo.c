/*
 * abi_demo.c -- libzkhip.so driven from PLAIN C (C99, no Python, no C++): what a compiled host -- the reference's Rust crates
 * through rust/zkhip_sys.rs -- does with include/zkhip.h.  Built by tests/native/Makefile (gcc), run by
 * tests/test_gpu_native_abi.py on the GPU box.  Every check compares two results that reach the host through DIFFERENT entry
 * points / kernels of the library, so the program needs no oracle:
 *   1  zk_msm_g1 is linear:  MSM(s1) + MSM(s2) == MSM(s1 + s2)            (zk_fr_add, zk_g1_lincomb with coefficients 1, 1)
 *   2  the window-table path (zk_srs_precompute) returns the same bits as the table-less path
 *   3  zk_msm_g1_batch == the single calls;  zk_msm_g1_batch_async + zk_msm_wait == zk_msm_g1_batch
 *   4  zk_open_rounds' value == zk_fold with every point                   (mode 3 vs mode 2 kernels, dpoly_comm.rs:309-323 / mle.rs:88-105)
 *   5  zk_sumcheck_batch == zk_sumcheck_product / zk_sumcheck / zk_open_rounds one call at a time
 *   6  one process, a ctx per visible GPU, zk_comm_init_all, one pthread per party (mpc-net/src/multi.rs:330-352): zk_d_msm with
 *      coefficients (1, 0, ..) returns party 0's local MSM on every party; zk_allgather moves every party's block
 *   7  error codes: ZK_ERR_LENGTH with the reference's Err(min_len), ZK_ERR_INVALID for a non-power-of-two table
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkhip.h"
#include "zkhip_test.h" /* (one test hook below: zk_dbg_tune) */

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != 0) {                                                                               \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, zk_last_error(ctx)); \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) {                                                        \
            fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

static uint64_t sm_state;
static uint64_t splitmix(void) {
    uint64_t z = (sm_state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
/* n field elements: any 4-limb pattern below r is the Montgomery form of some element; 254-bit values are below r */
static uint64_t *rand_fr(size_t n, uint64_t seed) {
    uint64_t *a = (uint64_t *)malloc(n * 32);
    sm_state = seed;
    for (size_t i = 0; i < 4 * n; i++) a[i] = splitmix();
    for (size_t i = 0; i < n; i++) a[4 * i + 3] &= 0x3fffffffffffffffULL;
    return a;
}
static void *to_device(zk_ctx *ctx, const void *h, size_t bytes) {
    void *d = NULL;
    CHECK(zk_malloc(ctx, bytes, &d));
    CHECK(zk_memcpy_h2d(ctx, d, h, bytes));
    return d;
}

struct party {
    zk_ctx *ctx;
    int id, world;
    uint64_t out[18], local[18];
    uint8_t *gathered;
    int ok;
};
static void *party_main(void *arg) {
    struct party *p = (struct party *)arg;
    zk_ctx *ctx = p->ctx;
    const size_t n = 1 << 10;
    zk_srs *srs = NULL;
    const uint64_t k0[4] = {77 + (uint64_t)p->id, 0, 0, 0}, k1[4] = {5, 0, 0, 0};
    CHECK(zk_srs_generate(ctx, k0, k1, n, &srs));
    uint64_t *s = rand_fr(n, 1000 + (uint64_t)p->id);
    void *d_s = to_device(ctx, s, n * 32);
    CHECK(zk_msm_g1(ctx, srs, 0, d_s, n, p->local));
    /* out = sum_i coeff_i * C_i with coeff = e_0: party 0's local result, on every party */
    uint64_t *coeff = (uint64_t *)calloc((size_t)p->world, 32);
    coeff[0] = 1;
    const zk_srs *srs_l[1] = {srs};
    const void *sc_l[1] = {d_s};
    const size_t n_l[1] = {n};
    CHECK(zk_d_msm(ctx, 1, srs_l, NULL, sc_l, n_l, NULL, coeff, p->out));
    /* every party's 64-byte block, ordered by party */
    uint8_t mine[64];
    memset(mine, 0x10 + p->id, sizeof(mine));
    void *d_mine = to_device(ctx, mine, 64), *d_all = NULL;
    CHECK(zk_malloc(ctx, 64 * (size_t)p->world, &d_all));
    CHECK(zk_allgather(ctx, d_mine, 64, d_all));
    p->gathered = (uint8_t *)malloc(64 * (size_t)p->world);
    CHECK(zk_memcpy_d2h(ctx, p->gathered, d_all, 64 * (size_t)p->world));
    zk_free(ctx, d_s), zk_free(ctx, d_mine), zk_free(ctx, d_all);
    zk_srs_free(ctx, srs);
    free(s), free(coeff);
    p->ok = 1;
    return NULL;
}

int main(void) {
    zk_ctx *ctx = NULL;
    if (zk_ctx_create(0, &ctx) != 0) {
        fprintf(stderr, "zk_ctx_create failed: no MI355X visible (the library has no CPU fallback)\n");
        return 2;
    }
    printf("%s, %d device(s)\n", zk_version(), zk_device_count());
    const size_t n = 1 << 14;
    const uint64_t k0[4] = {0x1234567, 0, 0, 0}, k1[4] = {0x89abcd, 0, 0, 0};
    zk_srs *srs = NULL;
    CHECK(zk_srs_generate(ctx, k0, k1, n, &srs));
    EXPECT(zk_srs_len(srs) == n);
    uint64_t *s1 = rand_fr(n, 1), *s2 = rand_fr(n, 2);
    void *d1 = to_device(ctx, s1, n * 32), *d2 = to_device(ctx, s2, n * 32), *d3 = NULL;
    CHECK(zk_malloc(ctx, n * 32, &d3));

    /* 1: linearity */
    uint64_t r1[18], r2[18], r3[18], sum[18], pts[36];
    CHECK(zk_msm_g1(ctx, srs, 0, d1, n, r1));
    CHECK(zk_msm_g1(ctx, srs, 0, d2, n, r2));
    CHECK(zk_fr_add(ctx, d1, d2, d3, n));
    CHECK(zk_msm_g1(ctx, srs, 0, d3, n, r3));
    memcpy(pts, r1, 144), memcpy(pts + 18, r2, 144);
    const uint64_t ones[8] = {1, 0, 0, 0, 1, 0, 0, 0};
    CHECK(zk_g1_lincomb(ctx, pts, ones, 2, sum));
    EXPECT(memcmp(sum, r3, 144) == 0);

    /* 2: window table, same bits */
    uint64_t r1t[18];
    CHECK(zk_srs_precompute(ctx, srs, 0));
    EXPECT(zk_srs_table_window(srs) > 0);
    CHECK(zk_msm_g1(ctx, srs, 0, d1, n, r1t));
    EXPECT(memcmp(r1t, r1, 144) == 0);

    /* 3: batch and asynchronous batch */
    const zk_srs *srs3[3] = {srs, srs, srs};
    const void *sc3[3] = {d1, d2, d3};
    const size_t n3[3] = {n, n, n / 4};
    uint64_t b[3 * 18], a[3 * 18], q4[18];
    CHECK(zk_msm_g1_batch(ctx, 3, srs3, NULL, sc3, n3, b));
    CHECK(zk_msm_g1(ctx, srs, 0, d3, n / 4, q4));
    EXPECT(memcmp(b, r1, 144) == 0 && memcmp(b + 18, r2, 144) == 0 && memcmp(b + 36, q4, 144) == 0);
    zk_msm_job *job = NULL;
    CHECK(zk_msm_g1_batch_async(ctx, 3, srs3, NULL, sc3, n3, &job));

    /* 4 + 5 run while the job is in flight */
    const size_t len = 1 << 12;
    uint64_t *chal = rand_fr(12, 3);
    void *d_q = NULL, *d_fold = NULL, *d_q2 = NULL;
    CHECK(zk_malloc(ctx, len * 32, &d_q));
    CHECK(zk_malloc(ctx, len * 32, &d_q2));
    CHECK(zk_malloc(ctx, 32, &d_fold));
    uint64_t value[4], folded[4];
    CHECK(zk_open_rounds(ctx, d1, len, chal, d_q, value));
    CHECK(zk_fold(ctx, d1, len, chal, 12, d_fold));
    CHECK(zk_memcpy_d2h(ctx, folded, d_fold, 32));
    EXPECT(memcmp(value, folded, 32) == 0);

    uint64_t tr[12 * 12], lf[4], lg[4], pr[12 * 8], last[4];
    CHECK(zk_sumcheck_product(ctx, d1, d2, len, chal, tr, lf, lg));
    CHECK(zk_sumcheck(ctx, d2, len, chal, pr, last));
    uint64_t btr[12 * 12], blf[4], blg[4], bpr[12 * 8], blast[4], bval[4];
    zk_sc_item items[3];
    memset(items, 0, sizeof(items));
    items[0].mode = 1, items[0].d_f = d1, items[0].d_g = d2, items[0].len = len, items[0].h_chal = chal, items[0].h_sums = btr, items[0].h_last_f = blf, items[0].h_last_g = blg;
    items[1].mode = 0, items[1].d_f = d2, items[1].len = len, items[1].h_chal = chal, items[1].h_sums = bpr, items[1].h_last_f = blast;
    items[2].mode = 3, items[2].d_f = d1, items[2].len = len, items[2].h_chal = chal, items[2].h_last_f = bval, items[2].d_out = d_q2;
    CHECK(zk_sumcheck_batch(ctx, 3, items));
    EXPECT(memcmp(btr, tr, sizeof(tr)) == 0 && memcmp(blf, lf, 32) == 0 && memcmp(blg, lg, 32) == 0);
    EXPECT(memcmp(bpr, pr, sizeof(pr)) == 0 && memcmp(blast, last, 32) == 0 && memcmp(bval, value, 32) == 0);
    uint64_t *hq = (uint64_t *)malloc(len * 32), *hq2 = (uint64_t *)malloc(len * 32);
    CHECK(zk_memcpy_d2h(ctx, hq, d_q, (len - 1) * 32));
    CHECK(zk_memcpy_d2h(ctx, hq2, d_q2, (len - 1) * 32));
    EXPECT(memcmp(hq, hq2, (len - 1) * 32) == 0);

    CHECK(zk_msm_wait(ctx, job, a));
    EXPECT(memcmp(a, b, sizeof(b)) == 0);

    /* 7: error behaviour */
    uint64_t tmp[18];
    size_t min_len = 0;
    uint64_t two_pts[24];
    EXPECT(zk_srs_download(ctx, srs, NULL) == ZK_ERR_INVALID);
    {
        uint64_t *all = (uint64_t *)malloc(n * 96);
        CHECK(zk_srs_download(ctx, srs, all));
        memcpy(two_pts, all, 192);
        free(all);
    }
    EXPECT(zk_msm_g1_host(ctx, two_pts, 96, 2, s1, 5, tmp, &min_len) == ZK_ERR_LENGTH && min_len == 2);
    EXPECT(zk_sumcheck(ctx, d1, 1000, chal, pr, last) == ZK_ERR_INVALID);
    EXPECT(zk_dbg_tune("no_such_knob", 1) == ZK_ERR_INVALID);

    /* 6: one process, a ctx per GPU, a thread per party */
    int world = zk_device_count();
    if (world > 8) world = 8;
    zk_ctx **ctxs = (zk_ctx **)calloc((size_t)world, sizeof(*ctxs));
    struct party *pa = (struct party *)calloc((size_t)world, sizeof(*pa));
    pthread_t *th = (pthread_t *)calloc((size_t)world, sizeof(*th));
    for (int p = 0; p < world; p++) EXPECT(zk_ctx_create(p, &ctxs[p]) == 0);
    EXPECT(zk_comm_init_all(ctxs, world) == 0);
    for (int p = 0; p < world; p++) {
        pa[p].ctx = ctxs[p], pa[p].id = p, pa[p].world = world;
        EXPECT(zk_comm_rank(ctxs[p]) == p && zk_comm_size(ctxs[p]) == world);
        pthread_create(&th[p], NULL, party_main, &pa[p]);
    }
    for (int p = 0; p < world; p++) pthread_join(th[p], NULL);
    for (int p = 0; p < world; p++) {
        EXPECT(pa[p].ok);
        EXPECT(memcmp(pa[p].out, pa[0].local, 144) == 0);
        for (int q = 0; q < world; q++) EXPECT(pa[p].gathered[64 * q] == 0x10 + q && pa[p].gathered[64 * q + 63] == 0x10 + q);
        zk_ctx_destroy(ctxs[p]);
    }

    zk_srs_free(ctx, srs);
    zk_ctx_destroy(ctx);
    printf("abi_demo ok: linearity, window table, batch, async, open == fold, sumcheck batch, errors, %d part%s over zk_comm_init_all\n", world, world == 1 ? "y" : "ies");
    return 0;
}
