// Go / no-go experiment of round 5: BATCHED-AFFINE bucket accumulation against the shipped XYZZ mixed addition
// (csrc/zk_msm.hip k_accum_tiles: one madd-2008-s = 8M + 2S = 9.04 multiplication-equivalents per sorted entry).
//
// An affine addition with a shared inversion is 5M + 1S (Montgomery's trick: 1M forward, 2M backward; lambda, lambda^2,
// lambda (x1 - x3)) = 5.77 multiplication-equivalents -- IF one inversion is shared by enough INDEPENDENT additions.  The additions
// of a bucket sum are independent only across buckets and across the pairs of one level of a pairwise tree, so the best case
// for the technique is what this file measures: the 14 x 2^20 sorted entries of the 2^20-point table-mode MSM (2^18 buckets,
// 56 entries each) reduced level by level, every lane owning K independent pairs per level,
//     forward   d_k = x2 - x1 of its K pairs, running product c, prefix c_k parked in HBM (coalesced [k][lane] layout)
//     inversion ONE per lane and level (f30_inv, Fermat; or NONE: `free`, an upper bound for any faster inversion)
//     backward  1 / d_k from the prefixes, the K affine sums, results parked as 96-byte points in the 64-point chunked layout
// Level 1 gathers its operands from a 1.4 GB table of 96-byte records through a random index list (the access pattern of the real
// sorted entries: every entry names one of 14 x 2^20 records), the upper levels stream the level below.  The same harness runs
// the shipped formula (xyzz30_madd over tiles of 32 gathered entries) as the reference point, and -- `check` -- compares the
// affine results with XYZZ chains normalised by a true inversion (the chord formulas are identities of rational functions: they
// agree on arbitrary coordinate pairs, on the curve or not, so the table holds random field elements).
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -o batch_affine batch_affine.hip
//   ./batch_affine [log2_lanes=17] [reps=5]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../scalable-collaborative-zksnark_amd/csrc/curve30.cuh"
using namespace zk;
#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static constexpr int kBlk = 256;

// affine point arrays of the tree levels: blocks of 64 points, 16-byte chunk k (0..5: x, x, x, y, y, y) side by side
__device__ __forceinline__ size_t aff_off(size_t idx, int k) { return (idx >> 6) * 6144 + (size_t)k * 1024 + (idx & 63) * 16; }
__device__ __forceinline__ size_t pre_off(size_t idx, int k) { return (idx >> 6) * 3072 + (size_t)k * 1024 + (idx & 63) * 16; }
template <size_t (*OFF)(size_t, int)>
__device__ __forceinline__ Fq30 ld3(const void* base, size_t idx, int k0) {
    const char* b = reinterpret_cast<const char*>(base);
    u32 w[12];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint4 v = *reinterpret_cast<const uint4*>(b + OFF(idx, k0 + i));
        w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
    }
    return f30_from_words(w);
}
template <size_t (*OFF)(size_t, int)>
__device__ __forceinline__ void st3(void* base, size_t idx, int k0, const Fq30& a) {
    char* b = reinterpret_cast<char*>(base);
    u32 w[12];
    f30_to_words(a, w);
#pragma unroll
    for (int i = 0; i < 3; i++) *reinterpret_cast<uint4*>(b + OFF(idx, k0 + i)) = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// operand `which` (0 / 1) of pair `pid`: level 1 = a gathered table record (sign bit: the negated point), upper levels = the level below
template <bool L1>
__device__ __forceinline__ Fq30 op_x(const void* tab, const u32* idx, const void* in, size_t pid, int which) {
    if (L1) return f30_load(tab, (size_t)(idx[2 * pid + which] & 0x7fffffffu) * 96);
    return ld3<aff_off>(in, 2 * pid + which, 0);
}
template <bool L1>
__device__ __forceinline__ Fq30 op_y(const void* tab, const u32* idx, const void* in, size_t pid, int which) {
    if (L1) {
        const u32 v = idx[2 * pid + which];
        const Fq30 y = f30_load(tab, (size_t)(v & 0x7fffffffu) * 96 + 48);
        return (v >> 31) ? f30_neg_canon(y) : y;
    }
    return ld3<aff_off>(in, 2 * pid + which, 3);
}

// value bounds: stored coordinates x < 6q, y < 4q (table records are canonical); every product has input bounds <= 256 q^2
// INV: 0 = one Fermat inversion per lane and level, 1 = none (the product stands in for its inverse: timing only)
template <bool L1, int INV>
static __global__ void __launch_bounds__(kBlk) k_tree_level(const void* __restrict__ tab, const u32* __restrict__ idx, const void* __restrict__ in,
                                                            size_t n_pairs, int K, size_t nl, void* __restrict__ out, void* __restrict__ pre) {
    const size_t g = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (g >= nl) return;
    Fq30 c = f30_one();
    for (int k = 0; k < K; k++) {  // forward: prefix products of the x differences
        const size_t pid = (size_t)k * nl + g;
        if (pid >= n_pairs) break;
        const Fq30 x1 = op_x<L1>(tab, idx, in, pid, 0), x2 = op_x<L1>(tab, idx, in, pid, 1);
        st3<pre_off>(pre, pid, 0, c);
        c = f30_mul(c, f30_sub8(x2, x1));  // 2q * 14q
    }
    Fq30 inv = INV == 0 ? f30_inv(c) : c;  // < 2q
    for (int k = K - 1; k >= 0; k--) {  // backward
        const size_t pid = (size_t)k * nl + g;
        if (pid >= n_pairs) continue;
        const Fq30 x1 = op_x<L1>(tab, idx, in, pid, 0), x2 = op_x<L1>(tab, idx, in, pid, 1);
        const Fq30 y1 = op_y<L1>(tab, idx, in, pid, 0), y2 = op_y<L1>(tab, idx, in, pid, 1);
        const Fq30 d = f30_sub8(x2, x1);                                // < 14q
        const Fq30 id = f30_mul(inv, ld3<pre_off>(pre, pid, 0));        // 1 / d_k  (< 2q)
        inv = f30_mul(inv, d);                                          // 2q * 14q
        const Fq30 lam = f30_mul(f30_sub8(y2, y1), id);                 // 12q * 2q
        Fq30 x3 = f30_sub12(f30_sqr(lam), f30_add(x1, x2));             // lam^2 + 12q - (x1 + x2) < 14q
        x3 = f30_csub_4q(f30_csub_4q(x3));                              // < 6q
        Fq30 y3 = f30_sub8(f30_mul(lam, f30_sub6(x1, x3)), y1);         // lam (x1 + 6q - x3) + 8q - y1 < 10q ... y1 < 4q: < 2q + 8q
        y3 = f30_csub_4q(f30_csub_4q(y3));                              // < 4q  (10q -> 6q -> < 4q holds only below 8q: second csub covers it)
        st3<aff_off>(out, pid, 0, x3);
        st3<aff_off>(out, pid, 3, y3);
    }
}

// the shipped accumulation formula on the same operands: a lane walks T gathered entries with one XYZZ accumulator
static __global__ void __launch_bounds__(kBlk) k_xyzz_tiles(const void* __restrict__ tab, const u32* __restrict__ idx, size_t n_tiles, int T, void* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * kBlk + threadIdx.x;
    if (g >= n_tiles) return;
    const u32* run = idx + g * T;
    Xyzz30 acc;
    xyzz30_set_inf(acc);
    u32 v = run[0];
    Aff30 p = aff30_load(tab, v & 0x7fffffffu);
    for (int e = 0; e < T; e++) {
        const bool neg = (v >> 31) != 0;
        const Aff30 cur = p;
        if (e + 1 < T) {
            v = run[e + 1];
            p = aff30_load(tab, v & 0x7fffffffu);
        }
        xyzz30_madd(acc, cur, neg);
    }
    xyzz30_store(out, g, acc);
}

// check: pair `pid` of level 1 through the XYZZ formulas and a true inversion; mismatches are counted
static __global__ void __launch_bounds__(64) k_check_level1(const void* __restrict__ tab, const u32* __restrict__ idx, const void* __restrict__ out, size_t n, u32* bad) {
    const size_t pid = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (pid >= n) return;
    Xyzz30 acc;
    xyzz30_set_inf(acc);
    for (int w = 0; w < 2; w++) {
        const u32 v = idx[2 * pid + w];
        xyzz30_madd(acc, aff30_load(tab, v & 0x7fffffffu), (v >> 31) != 0);
    }
    const Fq30 izz = f30_inv(acc.zz), izzz = f30_inv(acc.zzz);
    const Fq30 x = f30_canon8(f30_mul(acc.x, izz)), y = f30_canon8(f30_mul(acc.y, izzz));
    const Fq30 gx = f30_canon8(ld3<aff_off>(out, pid, 0)), gy = f30_canon8(ld3<aff_off>(out, pid, 3));
    u32 dx = 0, dy = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) dx |= x.l[i] ^ gx.l[i], dy |= y.l[i] ^ gy.l[i];
    if (dx | dy) {
        const u32 slot = atomicAdd(bad, 1u);
        if (slot < 8) {  // (pid, which coordinate, the two index words) of the first mismatches
            bad[1 + 4 * slot] = (u32)pid, bad[2 + 4 * slot] = (dx ? 1u : 0u) | (dy ? 2u : 0u);
            bad[3 + 4 * slot] = idx[2 * pid], bad[4 + 4 * slot] = idx[2 * pid + 1];
        }
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static __global__ void k_fill_table(uint32_t* tab, size_t nwords) {  // 12-word coordinates, top word < 2^28 (< q)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    uint32_t w = (uint32_t)mix64(i * 0x9e3779b97f4a7c15ull + 12345);
    if (i % 12 == 11) w &= 0x0fffffffu;
    tab[i] = w;
}
static __global__ void k_fill_idx(uint32_t* idx, size_t n, uint32_t nt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the two entries of a pair never name the same record: P + P / P - P (x2 == x1) is the exceptional case of the chord formula --
    // a zero difference would poison the whole batch of its lane; a production kernel would route such pairs to the XYZZ path
    const uint64_t r = mix64(i * 0xd1342543de82ef95ull + 777), r0 = mix64((i & ~(size_t)1) * 0xd1342543de82ef95ull + 777);
    uint32_t rec = (uint32_t)((r >> 1) % nt);
    if ((i & 1) && rec == (uint32_t)((r0 >> 1) % nt)) rec = (rec + 1) % nt;
    idx[i] = rec | ((uint32_t)(r & 1) << 31);
}

int main(int argc, char** argv) {
    const int lg_lanes = argc > 1 ? atoi(argv[1]) : 17, reps = argc > 2 ? atoi(argv[2]) : 5;
    const size_t NT = (size_t)14 << 20, E = (size_t)14 << 20, nl = (size_t)1 << lg_lanes;
    void *tab, *lvA, *lvB, *pre, *xout;
    u32 *idx, *bad;
    CHECK(hipMalloc(&tab, NT * 96));
    CHECK(hipMalloc(&idx, E * 4));
    CHECK(hipMalloc(&lvA, (E / 2 + 64) * 96));
    CHECK(hipMalloc(&lvB, (E / 4 + 64) * 96));
    CHECK(hipMalloc(&pre, (E / 2 + 64) * 48));
    CHECK(hipMalloc(&xout, (E / 32 + 64) * 192));
    CHECK(hipMalloc(&bad, 4 * 33));
    k_fill_table<<<(unsigned)((NT * 24 + 255) / 256), 256>>>((uint32_t*)tab, NT * 24);
    k_fill_idx<<<(unsigned)((E + 255) / 256), 256>>>(idx, E, (uint32_t)NT);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto ms_of = [&](auto&& fn) {
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps + 1; r++) {
            hipEventRecord(e0);
            fn();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r) best = ms < best ? ms : best, sum += ms;
        }
        return std::pair<float, float>(best, sum / reps);
    };
    printf("batched-affine gate: %zu sorted entries (14 x 2^20), table of %zu 96-byte records (%.2f GB), %zu lanes\n", E, NT, NT * 96 / 1e9, nl);

    // reference point: the shipped mixed addition, tiles of 32 entries (k_accum_tiles' inner loop on the same operands)
    {
        const int T = 32;
        const size_t tiles = E / T;
        auto r = ms_of([&] { k_xyzz_tiles<<<(unsigned)((tiles + kBlk - 1) / kBlk), kBlk>>>(tab, idx, tiles, T, xout); });
        printf("xyzz madd tiles T=32         : best %.3f ms  mean %.3f ms  -> %.2f G additions/s\n", r.first, r.second, E / (r.second * 1e-3) / 1e9);
    }
    for (int inv = 0; inv < 2; inv++) {
        float tot_best = 0, tot_mean = 0;
        size_t adds = 0;
        printf("affine tree, inversion = %s\n", inv == 0 ? "Fermat (f30_inv), one per lane and level" : "NONE (upper bound for any inversion)");
        size_t n_pairs = E / 2;
        void *src = nullptr, *dst = lvA;
        for (int level = 1; n_pairs >= ((size_t)1 << 17); level++) {
            const int K = (int)((n_pairs + nl - 1) / nl);
            const unsigned grid = (unsigned)((nl + kBlk - 1) / kBlk);
            std::pair<float, float> r;
            if (level == 1) {
                r = inv == 0 ? ms_of([&] { k_tree_level<true, 0><<<grid, kBlk>>>(tab, idx, nullptr, n_pairs, K, nl, dst, pre); })
                             : ms_of([&] { k_tree_level<true, 1><<<grid, kBlk>>>(tab, idx, nullptr, n_pairs, K, nl, dst, pre); });
            } else {
                r = inv == 0 ? ms_of([&] { k_tree_level<false, 0><<<grid, kBlk>>>(nullptr, nullptr, src, n_pairs, K, nl, dst, pre); })
                             : ms_of([&] { k_tree_level<false, 1><<<grid, kBlk>>>(nullptr, nullptr, src, n_pairs, K, nl, dst, pre); });
            }
            printf("  level %d: %8zu additions, K = %3d per lane : best %.3f ms  mean %.3f ms  -> %.2f G additions/s\n", level, n_pairs, K, r.first, r.second,
                   n_pairs / (r.second * 1e-3) / 1e9);
            if (level == 1 && inv == 0) {
                CHECK(hipMemset(bad, 0, 4 * 33));
                const size_t nchk = 1 << 20;
                k_check_level1<<<(unsigned)(nchk / 64), 64>>>(tab, idx, dst, nchk, bad);
                u32 hb[33];
                CHECK(hipMemcpy(hb, bad, 4 * 33, hipMemcpyDeviceToHost));
                printf("  check: %u of %zu level-1 sums differ from the XYZZ formulas normalised by a true inversion\n", hb[0], nchk);
                for (u32 i = 0; i < hb[0] && i < 8; i++)
                    printf("    pair %u (k = %zu, lane %zu): %s differ; entries %08x %08x\n", hb[1 + 4 * i], hb[1 + 4 * i] / nl, hb[1 + 4 * i] % nl,
                           hb[2 + 4 * i] == 1 ? "x" : hb[2 + 4 * i] == 2 ? "y" : "x and y", hb[3 + 4 * i], hb[4 + 4 * i]);
            }
            tot_best += r.first, tot_mean += r.second, adds += n_pairs;
            src = dst;
            dst = (dst == lvA) ? lvB : lvA;
            n_pairs /= 2;
        }
        printf("  whole tree: %zu additions in %.3f ms (mean; best %.3f) -> %.2f G additions/s\n", adds, tot_mean, tot_best, adds / (tot_mean * 1e-3) / 1e9);
    }
    return 0;
}
