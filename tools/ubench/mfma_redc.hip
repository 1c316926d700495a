// Experiment (round 3): can the two CONSTANT-operand products of a Montgomery multiplication
// (m = T_lo * q' mod R, then m * q) move to the idle MFMA pipe as v_mfma_i32_16x16x64_i8 over byte limbs, leaving only the
// variable x variable product on the VALU?
//
// A constant-operand product of a batch of elements is a GEMM: (Toeplitz matrix of the constant's bytes) x (bytes of the elements).
// For a 390-bit Fq element: 49 byte limbs.  m = low 49 byte columns of T_lo * q'  -> 4 M-tiles of 16 columns x 1 K-tile (49 <= 64);
// m * q needs all 98 columns -> 7 M-tiles.  11 MFMAs per 16 elements, 44 per wave of 64 elements.  The MFMA pipe takes them for free
// beside the VALU -- IF the VALU work they leave behind is smaller than the VALU work they replace.  What they leave behind:
//   * byte-split of the 13 x 30-bit limbs of T_lo into the B-operand layout, and of m again between the two GEMMs;
//   * the C/D tiles hold 8-bit-spaced column sums (49 + 98 = 147 i32 per element): carry propagation back into 13 x 30-bit limbs;
//   * cross-lane transposes between "one element per lane" and the MFMA operand / result maps (NOT counted below: optimistic).
// What they replace: 169 v_mad_u64_u32 + 13 v_mul_lo_u32 + the m masks (the reduction half of fq30.cuh f30_mul).
//
// Kernels timed (one element per lane, dependent chains, 2 waves per SIMD like the MSM accumulation):
//   full      f30_mul as shipped                                   (product + reduction on the VALU)
//   prod      the variable x variable product alone + its carries   (what stays on the VALU in any case)
//   epilogue  the MINIMUM VALU epilogue of the MFMA path for one element: 147 column sums -> bytes of m (repacked 4 per dword) -> 13 limbs
//   mfma      44 x v_mfma_i32_16x16x64_i8 per wave-multiplication   (the MFMA pipe's share)
//   prod+epilogue+mfma  all three in one loop: MFMA overlaps, the VALU parts add
// Build: hipcc -O3 --offload-arch=gfx950 -I scalable-collaborative-zksnark_amd/csrc tools/ubench/mfma_redc.hip -o tools/ubench/mfma_redc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fq30.cuh"
using namespace zk;

typedef int v4i __attribute__((ext_vector_type(4)));
static constexpr int kIters = 256;

__device__ __forceinline__ Fq30 load_seed(const u32* p) {
    Fq30 x;
#pragma unroll
    for (int i = 0; i < 13; i++) x.l[i] = p[threadIdx.x * 13 + i] & (i < 12 ? Q30::MASK : 0x3ffffu);
    return x;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_full(const u32* in, u32* out) {
    Fq30 x = load_seed(in), y = load_seed(in + 13 * 256);
    for (int it = 0; it < kIters; it++) x = f30_mul(x, y);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) o ^= x.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = o;
}

// the 25 product columns of a * b with their carry sweep (26 limbs), folded back to 13 so the chain stays dependent
__device__ __forceinline__ Fq30 prod_only(const Fq30& a, const Fq30& b) {
    u32 t[26];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (j >= 0 && j < 13) acc += (u64)a.l[i] * b.l[j];
        }
        t[k] = (u32)acc & Q30::MASK;
        acc >>= 30;
    }
    t[25] = (u32)acc;
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = (t[i] ^ t[13 + i]) & Q30::MASK;
    return r;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_prod(const u32* in, u32* out) {
    Fq30 x = load_seed(in), y = load_seed(in + 13 * 256);
    for (int it = 0; it < kIters; it++) x = prod_only(x, y);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) o ^= x.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = o;
}

// minimum VALU epilogue of the MFMA path for ONE element (optimistic: the 147 column sums are already in this lane):
//   49 column sums of T_lo*q' -> exact bytes of m (carry sweep, mask) -> 13 dwords of 4 packed bytes (the next GEMM's B operand)
//   98 column sums of m*q     -> carry sweep -> 13 limbs of 30 bits (the high half; the low half cancels against T_lo)
__device__ __forceinline__ Fq30 epilogue(const int (&cm)[49], const int (&cq)[98], u32 (&packed)[13]) {
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < 49; j++) {
        const u32 v = (u32)cm[j] + carry;
        const u32 byte = v & 0xffu;
        carry = v >> 8;
        if ((j & 3) == 0) packed[j >> 2] = byte;
        else packed[j >> 2] |= byte << (8 * (j & 3));
    }
    u64 c2 = 0;
    u32 bytes[49];
#pragma unroll
    for (int j = 0; j < 98; j++) {
        c2 += (u64)(u32)cq[j];
        if (j >= 49) bytes[j - 49] = (u32)c2 & 0xffu;
        c2 >>= 8;
    }
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; i++) {  // 30-bit limb i = bits [30 i, 30 i + 30) of the 49-byte integer
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 5; b++) {
            const int byte = (30 * i) / 8 + b;
            if (byte < 49) {
                const int sh = 8 * byte - 30 * i;
                v |= sh >= 0 ? (sh < 32 ? bytes[byte] << sh : 0u) : (bytes[byte] >> (-sh));
            }
        }
        r.l[i] = v & Q30::MASK;
    }
    return r;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_epilogue(const u32* in, u32* out) {
    Fq30 x = load_seed(in);
    int cm[49], cq[98];
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int j = 0; j < 49; j++) cm[j] = (int)(x.l[j % 13] >> (j % 7)) & 0xfffff;  // stand-ins for the MFMA results (20-bit column sums)
#pragma unroll
        for (int j = 0; j < 98; j++) cq[j] = (int)(x.l[(j + 5) % 13] >> (j % 5)) & 0xfffff;
        u32 packed[13];
        Fq30 r = epilogue(cm, cq, packed);
#pragma unroll
        for (int i = 0; i < 13; i++) x.l[i] = (r.l[i] ^ packed[i]) & Q30::MASK;
    }
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) o ^= x.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = o;
}

// the MFMA pipe's share: 44 instructions per wave-multiplication (4 + 7 M-tiles for each of the 4 N-tiles of 16 elements)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mfma(const u32* in, u32* out) {
    v4i a = {(int)in[threadIdx.x], (int)in[threadIdx.x + 1], (int)in[threadIdx.x + 2], (int)in[threadIdx.x + 3]};
    v4i b = {(int)in[threadIdx.x + 4], (int)in[threadIdx.x + 5], (int)in[threadIdx.x + 6], (int)in[threadIdx.x + 7]};
    v4i acc[11];
#pragma unroll
    for (int k = 0; k < 11; k++) acc[k] = v4i{0, 0, 0, 0};
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < 11; k++) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
    }
    int o = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) o ^= acc[k][0] ^ acc[k][1] ^ acc[k][2] ^ acc[k][3];
    out[blockIdx.x * 256 + threadIdx.x] = (u32)o;
}

// everything the MFMA path executes per multiplication, in one loop: the MFMAs overlap with the VALU, the VALU parts add up
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_path(const u32* in, u32* out) {
    Fq30 x = load_seed(in), y = load_seed(in + 13 * 256);
    v4i acc[11];
#pragma unroll
    for (int k = 0; k < 11; k++) acc[k] = v4i{0, 0, 0, 0};
    for (int it = 0; it < kIters; it++) {
        const Fq30 p = prod_only(x, y);
        v4i a = {(int)p.l[0], (int)p.l[1], (int)p.l[2], (int)p.l[3]}, b = {(int)p.l[4], (int)p.l[5], (int)p.l[6], (int)p.l[7]};
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < 11; k++) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
        int cm[49], cq[98];
#pragma unroll
        for (int j = 0; j < 49; j++) cm[j] = (acc[j % 11][j % 4] ^ (int)(p.l[j % 13] >> (j % 7))) & 0xfffff;
#pragma unroll
        for (int j = 0; j < 98; j++) cq[j] = (acc[(j + 3) % 11][(j + 1) % 4] ^ (int)(p.l[(j + 5) % 13] >> (j % 5))) & 0xfffff;
        u32 packed[13];
        const Fq30 r = epilogue(cm, cq, packed);
#pragma unroll
        for (int i = 0; i < 13; i++) x.l[i] = (r.l[i] ^ packed[i]) & Q30::MASK;
    }
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) o ^= x.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = o;
}

template <class K>
static double run(const char* name, K kern, const u32* d_in, u32* d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_in, d_out);
    hipDeviceSynchronize();
    double best = 1e9;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_in, d_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double per_s = (double)blocks * 256 * kIters / (best * 1e-3);
    printf("%-22s %8.3f ms   %7.2f G lane-iterations/s\n", name, best, per_s / 1e9);
    return per_s;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2 * 8;  // 2 workgroups of 4 waves per CU resident (2 waves / SIMD), 8 rounds
    std::vector<u32> h(26 * 256 + 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(0x9e3779b9u * (i + 1)) ^ (u32)(i << 7);
    u32 *d_in, *d_out;
    hipMalloc(&d_in, h.size() * 4);
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("%s, %d CUs, %d workgroups of 256 lanes, %d dependent iterations per lane\n", prop.gcnArchName, prop.multiProcessorCount, blocks, kIters);
    const double full = run("full f30_mul", k_full, d_in, d_out, blocks);
    const double prod = run("prod (VALU share)", k_prod, d_in, d_out, blocks);
    const double epi = run("epilogue (VALU, min.)", k_epilogue, d_in, d_out, blocks);
    const double mf = run("mfma x44 / wave-mul", k_mfma, d_in, d_out, blocks);
    const double path = run("prod+mfma+epilogue", k_path, d_in, d_out, blocks);
    printf("MFMA path / shipped multiplication: %.2fx the time (prod + epilogue alone, no MFMA: %.2fx)\n", full / path, full * (1.0 / prod + 1.0 / epi));
    (void)mf;
    return 0;
}
