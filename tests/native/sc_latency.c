/*
 * sc_latency.c -- wall time per call of the sumcheck family measured FROM C, i.e. at the C ABI itself (include/zkhip.h):
 * what a compiled host (the reference's Rust crates through rust/zkhip_sys.rs) pays per call, without the Python wrapper's
 * numpy allocations and ctypes conversions (tools/sc_time.py measures the wrapper).  Built by tests/native/Makefile.
 *   ./sc_latency [log2 sizes ...]          default 12 16 20 24
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zkhip.h"

#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        int rc_ = (call);                                                                                  \
        if (rc_ != 0) {                                                                                    \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, zk_last_error(ctx)); \
            exit(1);                                                                                       \
        }                                                                                                  \
    } while (0)

static uint64_t sm_state;
static uint64_t splitmix(void) {
    uint64_t z = (sm_state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
/* any 4-limb pattern below r is the Montgomery form of some element; 254-bit values are below r */
static uint64_t *rand_fr(size_t n, uint64_t seed) {
    uint64_t *a = (uint64_t *)malloc(n * 32);
    sm_state = seed;
    for (size_t i = 0; i < 4 * n; i++) a[i] = splitmix();
    for (size_t i = 0; i < n; i++) a[4 * i + 3] &= 0x3fffffffffffffffULL;
    return a;
}
static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int main(int argc, char **argv) {
    zk_ctx *ctx = NULL;
    int sizes[16], ns = 0;
    for (int i = 1; i < argc && ns < 16; i++) sizes[ns++] = atoi(argv[i]);
    if (ns == 0) sizes[0] = 12, sizes[1] = 16, sizes[2] = 20, sizes[3] = 24, ns = 4;
    if (zk_ctx_create(0, &ctx) != 0) {
        fprintf(stderr, "no GPU / libzkhip: zk_ctx_create failed\n");
        return 2;
    }
    for (int s = 0; s < ns; s++) {
        const int lg = sizes[s];
        const size_t n = (size_t)1 << lg;
        uint64_t *hf = rand_fr(n, 1), *hg = rand_fr(n, 2), *ch = rand_fr(64, 3);
        void *f = NULL, *g = NULL, *q = NULL, *out = NULL;
        CHECK(zk_malloc(ctx, n * 32, &f));
        CHECK(zk_malloc(ctx, n * 32, &g));
        CHECK(zk_malloc(ctx, n * 32, &q));
        CHECK(zk_malloc(ctx, 32, &out));
        CHECK(zk_memcpy_h2d(ctx, f, hf, n * 32));
        CHECK(zk_memcpy_h2d(ctx, g, hg, n * 32));
        uint64_t *sums = (uint64_t *)malloc((size_t)lg * 3 * 32), lf[4], lgv[4];
        const int reps = lg <= 22 ? 200 : 20;
        for (int mode = 0; mode < 4; mode++) {
            double best = 1e30, tot = 0;
            for (int it = -5; it < reps; it++) {
                const double t0 = now_us();
                if (mode == 0) CHECK(zk_sumcheck_product(ctx, f, g, n, ch, sums, lf, lgv));
                if (mode == 1) CHECK(zk_sumcheck(ctx, f, n, ch, sums, lf));
                if (mode == 2) {
                    CHECK(zk_fold(ctx, f, n, ch, (size_t)lg, out));
                    CHECK(zk_ctx_sync(ctx));
                }
                if (mode == 3) CHECK(zk_open_rounds(ctx, f, n, ch, q, lf));
                const double dt = now_us() - t0;
                if (it >= 0) {
                    tot += dt;
                    if (dt < best) best = dt;
                }
            }
            static const char *names[4] = {"product", "plain", "fold", "open"};
            static const int bytes[4] = {64, 32, 32, 64};
            printf("%-8s 2^%d: mean %9.1f us  min %9.1f us   %8.1f GB/s algorithmic (mean)   [C ABI, %d calls]\n", names[mode], lg, tot / reps, best,
                   (double)bytes[mode] * n / (tot / reps) / 1e3, reps);
        }
        free(hf), free(hg), free(ch), free(sums);
        zk_free(ctx, f), zk_free(ctx, g), zk_free(ctx, q), zk_free(ctx, out);
    }
    zk_ctx_destroy(ctx);
    return 0;
}
