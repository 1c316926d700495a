/*
 * zkhip.h -- C ABI of libzkhip.so: the MI355X (gfx950) drop-in for the compute loops of the
 * reference's `dist-primitive` crate (LBruyne/Scalable-Collaborative-zkSNARK).
 *
 * The reference has no FFI of its own (it is 100 % Rust on arkworks); these entry points are
 * what a `#[link(name = "zkhip")] extern "C"` block in dist-primitive would bind to replace
 *   - `G::msm(b, s)`                         dist-primitive/src/dmsm.rs:23, dpoly_comm.rs:242,274,457
 *   - the Phase-1 sumcheck loops             dsumcheck.rs:10-21,37-85,107-121,167-219,301-315,377-429
 *   - fold / fix_variable                    mle.rs:62-70,95-103
 *   - the open() quotient+fold loop          dpoly_comm.rs:309-323,337-351,418-432
 *   - the product tree                       dacc_product.rs:31-38,304-313,372-381
 *   - the element-wise Fr steps              hyperplonk/src/dhyperplonk.rs:233-238,251-256,326-339
 * INTEGRATION.md shows the Rust-side binding.
 *
 * Memory layouts are the reference's own, so a Rust caller passes its buffers unchanged:
 *   Fr  : 4 x u64 little-endian limbs, Montgomery form R = 2^256  (ark-ff Fp<MontBackend<_,4>,4>), 32 B
 *   Fq  : 6 x u64 limbs, Montgomery form R = 2^384, 48 B
 *   G1 affine: { x: Fq, y: Fq, infinity: bool }  -- stride 104 as a Rust struct; stride 96
 *              (x||y, with x = y = 0 meaning infinity) is accepted too
 *   G1 projective (results): Jacobian { x, y, z } 3 x 48 B = 144 B, as ark-ec Projective.
 *              Results are returned NORMALISED (z = R mod q, or (1,1,0) for infinity), so
 *              they are bit-comparable and are valid `Projective` values.
 *
 * Conventions: every function returns 0 on success or a negative zk_status; nothing aborts.
 * `d_` pointers are device (HBM) pointers, `h_` pointers are host pointers.  A ctx is bound
 * to one GPU; calls on distinct ctx are thread-safe, calls on one ctx are serialised by the
 * caller.  All work is enqueued on the ctx stream; functions that return host results
 * synchronise that stream before returning.
 */
#ifndef ZKHIP_H
#define ZKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;
typedef struct zk_srs zk_srs; /* device-resident base vector(s): one level of powers_of_g */

typedef enum {
    ZK_OK = 0,
    ZK_ERR_INVALID = -1,   /* bad argument (null, not a power of two, ...) : the reference's assert!s */
    ZK_ERR_LENGTH = -2,    /* bases/scalars length mismatch: `msm` -> Err(min_len); see zk_last_error */
    ZK_ERR_HIP = -3,       /* HIP runtime error */
    ZK_ERR_NO_DEVICE = -4, /* no gfx950 device / library built without device code */
    ZK_ERR_DIV_ZERO = -5,  /* zero denominator in zk_fr_batch_div (reference: inverse().unwrap() panic) */
    ZK_ERR_OOM = -6,
    ZK_ERR_COMM = -7       /* RCCL error / no communicator (reference: MPCNetError) */
} zk_status;

/* ---- context ------------------------------------------------------------------------ */
int zk_ctx_create(int device_id, zk_ctx **out);
void zk_ctx_destroy(zk_ctx *ctx);
const char *zk_last_error(zk_ctx *ctx);
/* use an externally owned hipStream_t (e.g. torch's current stream); NULL = ctx-owned stream */
int zk_ctx_set_stream(zk_ctx *ctx, void *hip_stream);
int zk_ctx_sync(zk_ctx *ctx);
const char *zk_version(void);
/* number of GPUs visible to the library (a party-per-GPU caller sizes its world with it); < 0: zk_status */
int zk_device_count(void);

/* free / total bytes of the ctx's GPU (hipMemGetInfo; parked zk_free blocks count as used) */
int zk_mem_info(zk_ctx *ctx, size_t *h_free, size_t *h_total);
/* The ARENA PLAN of a ctx: the sizes its scratch arenas (MSM passes, asynchronous MSM lanes, sumcheck family, pinned staging) have
 * grown to.  The arenas are sized on demand, so the FIRST proof of a process pays their allocation (n = 24: ~110 GB, seconds).  A
 * prover issues the same passes proof after proof: export the plan after one proof of a parameter-set shape (a few hundred bytes,
 * ZK_ARENA_PLAN_WORDS u64 -- keep it beside the proving key), import it right after zk_ctx_create in later processes, and the first
 * proof allocates nothing.  Import only grows arenas; ZK_ERR_INVALID for a buffer that is not a plan, ZK_ERR_OOM when the device
 * cannot hold it (what was allocated stays).  No reference counterpart: the reference allocates per call (Vec per round). */
#define ZK_ARENA_PLAN_WORDS 48
int zk_arena_plan_export(zk_ctx *ctx, uint64_t *h_plan /* [ZK_ARENA_PLAN_WORDS] */);
int zk_arena_plan_import(zk_ctx *ctx, const uint64_t *h_plan /* [ZK_ARENA_PLAN_WORDS] */);
/* ---- device memory helpers for non-HIP callers.  zk_free parks the block in the ctx (no device
 * synchronisation) and zk_malloc of the same size reuses it; everything is released with the ctx. -- */
int zk_malloc(zk_ctx *ctx, size_t bytes, void **d_out);
int zk_free(zk_ctx *ctx, void *d_ptr);
/* release every parked block to the driver now (other allocators in the process -- torch, RCCL -- cannot
 * see parked memory); *h_freed_bytes (optional) receives the number of bytes returned.  Every internal
 * allocation of the library does this by itself, and retries once, when the device is out of memory. */
int zk_trim(zk_ctx *ctx, size_t *h_freed_bytes);
int zk_memcpy_h2d(zk_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int zk_memcpy_d2h(zk_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int zk_memcpy_d2d(zk_ctx *ctx, void *d_dst, const void *d_src, size_t bytes); /* asynchronous, on the ctx stream */

/* ---- element-wise Fr (hyperplonk/src/dhyperplonk.rs:233-238,251-256,326-339) -------- */
int zk_fr_add(zk_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);
int zk_fr_sub(zk_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);
int zk_fr_mul(zk_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);
/* out[i] = a[i] + alpha*b[i] + beta          (`s + alpha*sid + beta`, dhyperplonk.rs:326-337);
 * d_a may be NULL (taken as zero): out = alpha*b + beta */
int zk_fr_axpb(zk_ctx *ctx, const void *d_a, const void *d_b, const uint64_t h_alpha[4],
               const uint64_t h_beta[4], void *d_out, size_t n);
/* A small PUBLIC Fr matrix applied to k vectors at once -- the PSS maps on field elements
 * (pack_from_public / unpack / unpack2, secret-sharing/src/pss.rs:93-171; degree reduction
 * degree_reduce.rs:17-23) for any packing factor l:
 *   out[j*out_vec_stride + r*out_row_stride] = sum_c M[r*cols + c] * in[j*in_vec_stride + c*in_comp_stride]
 * strides in elements; h_matrix = rows*cols Fr (Montgomery) on the host. */
int zk_fr_apply_matrix(zk_ctx *ctx, const uint64_t *h_matrix, size_t rows, size_t cols, const void *d_in,
                       size_t in_vec_stride, size_t in_comp_stride, void *d_out, size_t out_vec_stride,
                       size_t out_row_stride, size_t k);
/* The same maps by transforms, as the reference computes them (ark-poly radix-2 (coset) FFTs, pss.rs:93-171),
 * for packing factors where the dense matrix is the slower form (8l >= 64): every vector is interpolated on a
 * domain of size A, cut / zero-extended to min(A, B) coefficients, and evaluated on a domain of size B:
 *   c = IDFT_A(in[0..n_in) zero-padded);  c[i] *= scale[i];  e = DFT_B(c zero-padded);  out[r] = e[r * step], r < take.
 * h_winv: A/2 powers of omega_A^-1, h_w: B/2 powers of omega_B, h_scale: min(A, B) factors A^-1 (offB/offA)^i
 * (the coset offsets of `get_coset`), all Montgomery Fr; A, B powers of two <= 512.  Strides as above.
 *   pack_from_public: A = 2l (coset g), B = 8l, n_in = l (or 2l), take = 8l
 *   unpack:  A = 8l, B = 2l (coset g), take = l          unpack2: A = 8l, B = 4l (coset g), take = l, step = 2 */
int zk_fr_ntt_map(zk_ctx *ctx, size_t A, const uint64_t *h_winv, size_t B, const uint64_t *h_w, const uint64_t *h_scale,
                  size_t n_in, size_t take, size_t step, const void *d_in, size_t in_vec_stride, size_t in_comp_stride,
                  void *d_out, size_t out_vec_stride, size_t out_row_stride, size_t k);
/* strided views of the product tree (dacc_product.rs:41-55, dhyperplonk.rs:344-359):
 * even[i] = t[2i] (v(x,0)), odd[i] = t[2i+1] (v(x,1)), i < n; v(1,x) is the contiguous upper half. */
int zk_fr_deinterleave(zk_ctx *ctx, const void *d_t, void *d_even, void *d_odd, size_t n);
/* out[i] = num[i] / den[i] (dhyperplonk.rs:339), batched inversion; ZK_ERR_DIV_ZERO if a den is 0 */
int zk_fr_batch_div(zk_ctx *ctx, const void *d_num, const void *d_den, void *d_out, size_t n);

/* ---- sumcheck family ---------------------------------------------------------------- */
/* Phase-1 loop of sumcheck / c_sumcheck / d_sumcheck (dsumcheck.rs:10-21 = :107-121 = :301-315).
 * d_tab: len = 2^n Fr (not modified). h_chal: n Fr. h_out_pairs: n pairs (sum_lo, sum_hi) = 2n Fr.
 * h_last: the single remaining table element (the caller appends (0,last), runs pss2ss, ...). */
int zk_sumcheck(zk_ctx *ctx, const void *d_tab, size_t len, const uint64_t *h_chal,
                uint64_t *h_out_pairs, uint64_t h_last[4]);
/* Phase-1 loop of sumcheck_product / c_ / d_ (dsumcheck.rs:37-85 = :167-219 = :377-429).
 * h_out_triples: n x (t0,t1,t2) = 3n Fr; h_last_f / h_last_g: remaining elements. */
int zk_sumcheck_product(zk_ctx *ctx, const void *d_f, const void *d_g, size_t len,
                        const uint64_t *h_chal, uint64_t *h_out_triples, uint64_t h_last_f[4],
                        uint64_t h_last_g[4]);
/* fix_variable (mle.rs:88-105): fold min(n, n_points) times; d_out receives len >> rounds Fr. */
int zk_fold(zk_ctx *ctx, const void *d_tab, size_t len, const uint64_t *h_points, size_t n_points,
            void *d_out);
/* Phase 1 of open / d_local_open / c_open (dpoly_comm.rs:309-323 = :337-351 = :418-432):
 * for every round q_i = hi - lo then fold with point[i].  d_q_out receives len-1 Fr: q_0 (len/2)
 * followed by q_1 (len/4) ... q_{n-1} (1) -- exactly the scalar vectors of the n commitments.
 * h_value: the final evaluation. */
int zk_open_rounds(zk_ctx *ctx, const void *d_tab, size_t len, const uint64_t *h_point,
                   void *d_q_out, uint64_t h_value[4]);
/* product tree of acc_product / d_acc_product / c_acc_product (dacc_product.rs:31-38):
 * d_tree receives 2N Fr: tree[0..N) = x, tree[N+j] = tree[2j]*tree[2j+1], tree[2N-1] = 0. */
int zk_product_tree(zk_ctx *ctx, const void *d_x, size_t N, void *d_tree);

/* Several INDEPENDENT calls of the four functions above in one go.  A proof issues them in groups that do not depend on each
 * other -- three product sumchecks and three opens per layer of the wiring identity (hyperplonk/src/dhyperplonk.rs:417-478),
 * the six gate sumchecks (:223-260), the opens of :383-407 -- and most of them are chains of one to three latency-bound
 * launches.  The batch gives every item its own scratch, spreads the items over several streams and waits ONCE; every
 * output is bit-identical to the one-call-at-a-time form (same kernels, same launch plan per item).
 *   mode 0: zk_sumcheck          h_sums 2 log2(len) Fr, h_last_f = the remaining element
 *   mode 1: zk_sumcheck_product  h_sums 3 log2(len) Fr, h_last_f / h_last_g
 *   mode 2: zk_fold              n_points points, d_out receives len >> min(log2 len, n_points) Fr
 *   mode 3: zk_open_rounds       d_out receives the len - 1 quotient elements, h_last_f = the value */
typedef struct {
    int mode;
    const void *d_f;
    const void *d_g;
    size_t len;
    const uint64_t *h_chal;
    size_t n_points;
    uint64_t *h_sums;
    uint64_t *h_last_f;
    uint64_t *h_last_g;
    void *d_out;
} zk_sc_item;
int zk_sumcheck_batch(zk_ctx *ctx, size_t count, const zk_sc_item *items);

/* ---- G1 MSM -------------------------------------------------------------------------- */
/* Upload a base vector once (the reference clones powers_of_g[level] per call, dpoly_comm.rs:258).
 * h_bases: n affine points at `stride` bytes (96 or 104).  The device copy is the library's own:
 * packed at 96 B in its internal Montgomery form, followed by the endomorphism images (beta*x, y) of
 * every point (2 * n * 96 bytes).  Bases must lie in the prime-order subgroup, as arkworks' G1Affine
 * guarantees: the MSM splits scalars as k1 + k2*lambda and phi = [lambda] holds only there. */
int zk_srs_register(zk_ctx *ctx, const void *h_bases, size_t stride, size_t n, zk_srs **out);
/* Same from a device buffer in the packed 96-B reference layout.  The library keeps its own copy in
 * its internal Montgomery form (one conversion pass); the caller's buffer is not referenced afterwards. */
int zk_srs_wrap_device(zk_ctx *ctx, const void *d_bases96, size_t n, zk_srs **out);
/* Synthetic SRS on device: P_i = (k0 + i*k1)*G, the generator G1; mirrors the random-point SRS of
 * PolynomialCommitmentCub::new_single/new_random (dpoly_comm.rs:197-233). k0,k1 canonical 4xu64. */
int zk_srs_generate(zk_ctx *ctx, const uint64_t h_k0[4], const uint64_t h_k1[4], size_t n, zk_srs **out);
/* Structured SRS, PolynomialCommitmentCub::new (dpoly_comm.rs:37-67): out_levels[k], k = 0..nvars, receives
 * powers_of_g[k] = g^{E_k[j]}, E_0 = [1], E_{k+1} = E_k (1 - s_{nvars-k-1}) ++ E_k s_{nvars-k-1}  (2^k points).
 * h_g96: the base in the reference affine layout (NULL: the G1 generator); h_s: nvars Fr (Montgomery).
 * The exponents are expanded in Fr on the device and every point is one fixed-base multiplication. */
int zk_srs_powers(zk_ctx *ctx, const void *h_g96, const uint64_t *h_s, size_t nvars, zk_srs **out_levels);
/* One party's packed level, PolynomialCommitmentCub::to_packed (dpoly_comm.rs:164-194): out[k] = sum_{j<l}
 * row[j] * level[k l + j] with row = this party's row of the pack_from_public matrix (l CANONICAL 4 x u64
 * scalars); a level shorter than l is zero-extended to one chunk. */
int zk_srs_to_packed(zk_ctx *ctx, const zk_srs *level, const uint64_t *h_row, size_t l, zk_srs **out);
/* zk_fr_apply_matrix on G1 points -- the PSS maps are generic over DomainCoeff (pss.rs:93-171; the leader's
 * unpack2 / pack_from_public on commitments, dmsm.rs:30-39): device buffers of affine points in the reference
 * layout (96 B, x = y = 0: infinity), h_matrix = rows*cols CANONICAL 4 x u64 scalars,
 *   out[j*out_vec_stride + r*out_row_stride] = sum_c M[r*cols + c] * in[j*in_vec_stride + c*in_comp_stride]. */
int zk_g1_apply_matrix(zk_ctx *ctx, const uint64_t *h_matrix, size_t rows, size_t cols, const void *d_in96,
                       size_t in_vec_stride, size_t in_comp_stride, void *d_out96, size_t out_vec_stride,
                       size_t out_row_stride, size_t k);
/* Optional, once per SRS level (setup, like uploading it): build the table 2^{o_w} * P_i for every
 * window offset o_w of a `window_bits`-wide signed-digit decomposition (0 = pick for the level's
 * length).  MSMs on this SRS then use ONE bucket set for all windows: 1/W of the bucket-reduction
 * and fix-up work and no cross-window doubling chain.  Costs W x the level's memory (W ~ 13-22 copies).  A G1 table's
 * records are padded from 96 to 128 bytes -- one per cache line: the accumulation's gathers then move one line each, 2^20 MSM
 * +4-7 %, 4/3 of the table memory -- when that leaves at least 60 % of the device free at the moment of the call, and packed
 * (96 bytes) otherwise; zk_srs_precompute_layout forces one form.  window_bits <= 22. */
int zk_srs_precompute(zk_ctx *ctx, zk_srs *srs, int window_bits);
/* The same with the record layout of a G1 table chosen by the caller: record_bytes = 96 (packed), 128 (one record per 128-byte
 * cache line) or 0 (zk_srs_precompute's own choice, by free memory).  Same MSM results.  G2 levels ignore it (192-byte
 * records).  ZK_ERR_INVALID for any other value. */
int zk_srs_precompute_layout(zk_ctx *ctx, zk_srs *srs, int window_bits, int record_bytes);
/* bytes per record of the level's table (0: none built) */
int zk_srs_table_record(const zk_srs *srs);
/* window bits of the level's table (0: none built) -- the MSMs on it insert ceil(256 / bits) digits per scalar */
int zk_srs_table_window(const zk_srs *srs);
int zk_srs_free(zk_ctx *ctx, zk_srs *srs);
size_t zk_srs_len(const zk_srs *srs);
/* the library's device copy (INTERNAL Montgomery form, radix 2^390): for diagnostics only */
const void *zk_srs_device_ptr(const zk_srs *srs);
/* read the bases back in the reference layout (96 B per point, Montgomery radix 2^384) */
int zk_srs_download(zk_ctx *ctx, const zk_srs *srs, void *h_out96);

/* sum_i scalars[i] * bases[offset + i], i < n.  d_scalars: n Fr (Montgomery, as the reference
 * passes them; converted on device like `into_bigint`).  h_out: 18 u64 normalised Jacobian. */
int zk_msm_g1(zk_ctx *ctx, const zk_srs *srs, size_t offset, const void *d_scalars, size_t n,
              uint64_t h_out[18]);
/* A batch of independent MSMs in one pipeline pass -- the shape of d_msm's input
 * (`bases: &Vec<Vec<Affine>>, scalars: &Vec<Vec<Fr>>`, dmsm.rs:9-24; c_open hands it n+2 MSMs of
 * halving size, dpoly_comm.rs:435-436).  Items are grouped by window width and share the launch
 * sequence; all host combines run after ONE synchronisation, in parallel host threads.
 * offsets may be NULL (all zero).  h_out: count x 18 u64. */
int zk_msm_g1_batch(zk_ctx *ctx, size_t count, const zk_srs *const *srs, const size_t *offsets,
                    const void *const *d_scalars, const size_t *n, uint64_t *h_out);
/* Asynchronous form of zk_msm_g1_batch.  Nothing on the reference's path consumes an MSM result on the device (challenges are
 * pre-sampled, hyperplonk/src/dhyperplonk.rs:159-571), so a caller can start the commitments of a protocol step, run the
 * step's sumchecks on the same ctx while they are in flight, and collect the points afterwards.  The job is enqueued on
 * streams of its own (ordered after the work already enqueued on the ctx stream, which produced its scalars) with buffers of
 * its own; consecutive jobs alternate between two sets of streams, so the latency-bound end of one job overlaps with the
 * sort and the accumulation of the next.  The scalar buffers must stay valid and unmodified until zk_msm_wait, which runs the
 * host part, writes count x 18 u64 to h_out and releases the job (also on error). */
typedef struct zk_msm_job zk_msm_job;
int zk_msm_g1_batch_async(zk_ctx *ctx, size_t count, const zk_srs *const *srs, const size_t *offsets,
                          const void *const *d_scalars, const size_t *n, zk_msm_job **job);
int zk_msm_wait(zk_ctx *ctx, zk_msm_job *job, uint64_t *h_out);
/* ---- G2: `d_msm` / `G::msm` are generic over CurveGroup (dmsm.rs:9,23); the reference's parameters carry G2 points
 * in powers_of_g2 (dpoly_comm.rs:27,59-62).  Same pipeline, coordinates in Fq2 = Fq[u]/(u^2 + 1).
 * Layouts (ark-bls12-381): G2Affine = { x: Fq2{c0, c1}, y: Fq2, infinity } -> 192-byte records x.c0|x.c1|y.c0|y.c1,
 * stride 192 or the Rust struct's stride (flag byte at offset 192); results G2Projective = 3 x Fq2 = 36 u64, normalised.
 * zk_srs_download on a G2 vector writes 192-byte records; zk_srs_len / zk_srs_free apply unchanged. */
int zk_srs_register_g2(zk_ctx *ctx, const void *h_bases, size_t stride, size_t n, zk_srs **out);
int zk_msm_g2(zk_ctx *ctx, const zk_srs *srs, size_t offset, const void *d_scalars, size_t n, uint64_t h_out[36]);
int zk_msm_g2_batch(zk_ctx *ctx, size_t count, const zk_srs *const *srs, const size_t *offsets,
                    const void *const *d_scalars, const size_t *n, uint64_t *h_out);
/* Drop-in for `G::msm(&[Affine], &[Fr]) -> Result<G, usize>` on host slices (dmsm.rs:23):
 * returns ZK_ERR_LENGTH when n_bases != n_scalars and stores min(n_bases, n_scalars) in *h_err_len. */
int zk_msm_g1_host(zk_ctx *ctx, const void *h_bases, size_t stride, size_t n_bases,
                   const uint64_t *h_scalars, size_t n_scalars, uint64_t h_out[18], size_t *h_err_len);
/* K9 -- the leader's small public linear maps on points (d_msm closure dmsm.rs:30-39: unpack2 ->
 * sum -> pack_from_public; d_commit/d_open sums dpoly_comm.rs:289-292,372-391): sum_i k_i * P_i
 * for a handful of points.  h_points: n Jacobian points (18 u64 each, any representative),
 * h_scalars: n CANONICAL (non-Montgomery) 4xu64 scalars; h_out normalised Jacobian.  Runs on
 * the host: n is N_p = 8l and the work is one ~255-step dependency chain. */
int zk_g1_lincomb(zk_ctx *ctx, const uint64_t *h_points, const uint64_t *h_scalars, size_t n,
                  uint64_t h_out[18]);
/* `count` combinations with one shared scalar vector: out[r] = sum_i k_i * P[r*n + i]
 * (the per-proof-element sums of d_open, dpoly_comm.rs:372-376); inversions are batched. */
int zk_g1_lincomb_batch(zk_ctx *ctx, const uint64_t *h_points, const uint64_t *h_scalars, size_t n,
                        size_t count, uint64_t *h_out);
/* window size (bits) the device Pippenger picks for n points (2n entries of 128-bit half scalars per
 * window, ceil(129 / bits) windows); 0 < override <= 20 forces it */
int zk_msm_window(size_t n);
int zk_msm_set_window(zk_ctx *ctx, int c_override);
/* per-phase time of the last zk_msm_g1 (first window class of a batch) on this ctx, in ms, HIP
 * events on the ctx stream: [0] digits+sort, [1] k_accum_tiles (bucket accumulation kernel alone),
 * [2] fix-up, [3] bucket reduction + conversion + D2H, [4] host combine (wall), [5] total */
int zk_msm_last_timing(zk_ctx *ctx, float h_ms[6]);

/* device time of the last zk_sumcheck / zk_sumcheck_product / zk_open_rounds call on this ctx, HIP events on the ctx stream,
 * recorded only while the knob "sc_ts" is 3 (ZKHIP_TUNE=sc_ts=3, or zk_dbg_tune of include/zkhip_test.h): [0] the first stage (the first HBM pass of a large table:
 * k_pass<2,1> for the product sumcheck), [1] all launches of the call */
int zk_sumcheck_last_timing(zk_ctx *ctx, float h_ms[2]);

/* ---- party exchanges on one node: an RCCL communicator inside the ctx -------------------------
 * Replaces the typed adapter over mpc-net's TCP star, dist-primitive/src/utils/serializing_net.rs:11-141.
 * The party axis is the GPU axis (party p = rank p); payloads are raw Montgomery limbs in HBM, moved over
 * xGMI with no serialisation or compression; every call is enqueued on the ctx stream (zk_ctx_sync, or any
 * function returning host results, completes it).  RCCL is loaded at run time (librccl.so.1).
 *   serializing_net.rs pattern                               here
 *   worker_send_or_leader_receive_element          :11-39    zk_gather(root = 0)
 *   dynamic_worker_send_or_leader_receive_element  :41-72    zk_gather(root = receiver)
 *   worker_receive_or_leader_send_element          :74-96    zk_scatter(root = 0)
 *   dynamic_worker_receive_or_worker_send_element  :98-122   zk_scatter(root = sender)
 *   leader_compute_element                         :128-141  zk_allgather + the public map on every party
 *   loops of dynamic scatters over every root (dacc_product.rs:94-104,155-203)   zk_alltoall            */
#define ZK_COMM_ID_BYTES 128
/* rank 0 creates the id and hands it to the other parties out of band (once, over any channel) */
int zk_comm_unique_id(uint8_t h_id[ZK_COMM_ID_BYTES]);
/* one process per GPU: collective over all `world` parties */
int zk_comm_init(zk_ctx *ctx, int rank, int world, const uint8_t h_id[ZK_COMM_ID_BYTES]);
/* one process holding a ctx per GPU (the reference's model: one task per party, mpc-net/src/multi.rs:330-352):
 * ctxs[p] becomes party p.  The collectives below block until every party has entered them, so each party's calls
 * must come from ITS OWN host thread (as the reference's parties are separate tasks): driving two parties of one
 * communicator from a single thread deadlocks in the first exchange that returns host results (e.g. zk_d_msm). */
int zk_comm_init_all(zk_ctx *const *ctxs, int world);
int zk_comm_destroy(zk_ctx *ctx); /* also done by zk_ctx_destroy */
/* A party that cannot go on for a reason of its own (an out-of-memory arena, a failed kernel: anything outside the exchanges, which carry
 * their own status words) aborts its communicator (ncclCommAbort) so that its peers' pending and later collectives end with
 * ZK_ERR_COMM instead of waiting for it -- the library's form of the reference's `unwrap()` panic taking the job down
 * (mpc-net/src/multi.rs:330-352).  The ctx is left without a communicator. */
int zk_comm_abort(zk_ctx *ctx);
int zk_comm_rank(const zk_ctx *ctx);
int zk_comm_size(const zk_ctx *ctx);
/* d_recv[p * bytes ..] = party p's d_send, for every p, on every party */
int zk_allgather(zk_ctx *ctx, const void *d_send, size_t bytes, void *d_recv);
/* d_recv[p * bytes ..] = what party p put at d_send[me * bytes ..] */
int zk_alltoall(zk_ctx *ctx, const void *d_send, size_t bytes_per_peer, void *d_recv);
/* root receives world x bytes ordered by party (d_recv is ignored elsewhere) */
int zk_gather(zk_ctx *ctx, const void *d_send, size_t bytes, int root, void *d_recv);
/* party p receives d_send[p * bytes ..] of the root (d_send is ignored elsewhere) */
int zk_scatter(zk_ctx *ctx, const void *d_send, size_t bytes, int root, void *d_recv);
/* d_msm end to end (dmsm.rs:9-43): the batch of local MSMs (:19-24), the gather of the 144-byte results
 * (:29) as ONE all-gather, and the leader closure unpack2 -> sum -> pack_from_public (:30-39) as the public
 * linear map  h_out[k] = sum_i coeffs[i] * C_{i,k}  computed by every party for its own slot.
 * h_coeffs: world x 4 u64 CANONICAL scalars (for party p: coeffs[i] = c_p * lambda_i).  h_lambda (optional,
 * Montgomery): this party's scalars are multiplied by it on the device before its MSM; callers that pass
 * lambda_p = sum_j unpack2[j][p] pass coeffs[i] = c_p for all i (7 additions + one scalar multiplication).
 * h_out: count x 18 u64 normalised Jacobian -- this party's share of every result.
 * Error behaviour: a party whose local MSMs fail (ZK_ERR_LENGTH, ZK_ERR_OOM, ...) still takes part in the exchange and
 * returns its own error; every other party returns ZK_ERR_COMM naming the failed party.  The staging memory of the
 * exchange is taken before the local MSMs, so an out-of-memory party can still publish its status.  A party does NOT
 * join (its peers wait until its communicator is destroyed) only when: it has no communicator / an empty batch; the
 * few-KiB staging allocation itself fails; the HIP runtime or RCCL fails while enqueueing the exchange.
 * The 144-byte results travel pinned host -> device -> all-gather -> pinned host (the last steps of an MSM run on the
 * host, so the points exist there first): two PCIe hops of count x 144 x world bytes per call, see INTEGRATION.md. */
int zk_d_msm(zk_ctx *ctx, size_t count, const zk_srs *const *srs, const size_t *offsets, const void *const *d_scalars,
             const size_t *n, const uint64_t *h_lambda, const uint64_t *h_coeffs, uint64_t *h_out);

/* (the zk_dbg_* test hooks of the library are declared in include/zkhip_test.h: they are not part of this surface) */

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
