// Calibration of rocprofv3's FETCH_SIZE for the bucket kernel's access pattern: every lane gathers
// 96-byte records (six 16-byte loads) at random indices of a table far larger than the 256 MiB Infinity
// Cache, plus a coalesced 16 B/lane streaming read for reference.  Known byte counts are printed; run
// under `rocprofv3 --pmc FETCH_SIZE` and compare (tools/profile_bench.sh does, into profiles/).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_gather96(const uint4* __restrict__ tab, size_t nrec, unsigned* out, int iters) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = tid * 0x9E3779B97F4A7C15ull + 1;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const size_t r = s % nrec;
        const uint4* p = tab + r * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) { uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    out[tid] = acc;
}
__global__ void k_stream16(const uint4* __restrict__ tab, size_t n16, unsigned* out) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = tid; i < n16; i += stride) { uint4 v = tab[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    out[tid] = acc;
}
int main() {
    const size_t bytes = (size_t)3 << 30;  // 3 GiB table
    uint4* tab; unsigned* out;
    hipMalloc(&tab, bytes); hipMemset(tab, 1, bytes); hipMalloc(&out, (size_t)4 << 20);
    const size_t nrec = bytes / 96;
    const int blocks = 1024, threads = 256, iters = 64;
    k_gather96<<<blocks, threads>>>(tab, nrec, out, iters);
    k_stream16<<<blocks, threads>>>(tab, bytes / 16, out);
    hipDeviceSynchronize();
    const double gathers = (double)blocks * threads * iters;
    printf("k_gather96: %.0f gathers, requested %.1f KB, whole 128-B lines touched <= %.1f KB (1.75 lines per unaligned 96-B record on average)\n",
           gathers, gathers * 96 / 1024, gathers * 1.75 * 128 / 1024);
    printf("k_stream16: requested %.1f KB\n", (double)bytes / 1024);
    return 0;
}
