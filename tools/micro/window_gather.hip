// Micro-benchmark: random byte / 4-byte reads and 4-byte writes confined to a window of W bytes - what the memory system gives a
// kernel whose every access is a different line (the key gathers of the text rounds, the induction's character fetches, k_ibwt_walk,
// k_lcp; the rank array of the switch to prefix doubling for the writes).  Measurement infrastructure, not product code.
//   stand-alone:  hipcc --offload-arch=gfx950 -O3 tools/micro/window_gather.hip -o /tmp/wg && /tmp/wg      (profiles/r05_window_gather.txt)
//   as a library: hipcc ... -DPROBE_LIBRARY -fPIC -shared -o build/librandom_access_probe.so                (bench.py: "random_access")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill_idx(uint32_t* idx, uint64_t n, uint64_t mask, uint64_t seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        idx[i] = (uint32_t)(x & mask);
    }
}
__global__ void k_gather_u8(const uint32_t* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = src[__builtin_nontemporal_load(idx + i)];
}
__global__ void k_gather_u32(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ src, uint32_t* __restrict__ out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = src[__builtin_nontemporal_load(idx + i) >> 2];
}
__global__ void k_scatter_u32(const uint32_t* __restrict__ idx, uint32_t* __restrict__ dst, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[__builtin_nontemporal_load(idx + i) >> 2] = (uint32_t)i;
}

// mode 0: byte reads, 1: 4-byte reads, 2: 4-byte writes; window_bytes a power of two <= 4 GiB; best of `reps` launches of `accesses`
// accesses each, in G accesses / s.  Returns 0 or 1 (a HIP call failed: stderr says which).
extern "C" int msufsort_probe_random_access(int device, uint64_t window_bytes, uint64_t accesses, int mode, int reps, double* g_per_s)
{
    if (!g_per_s || mode < 0 || mode > 2 || reps < 1 || window_bytes < 4096 || (window_bytes & (window_bytes - 1)) || window_bytes > (4ull << 30) || accesses < 1) return 1;
    CK(hipSetDevice(device));
    uint32_t *idx = nullptr, *out = nullptr; uint8_t* win = nullptr;
    CK(hipMalloc(&idx, accesses * 4)); CK(hipMalloc(&out, accesses * 4)); CK(hipMalloc(&win, window_bytes));
    CK(hipMemset(win, 1, window_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k_fill_idx, dim3(grid), dim3(256), 0, 0, idx, accesses, window_bytes - 1, window_bytes >> 20);
    float best = 0;
    for (int rep = 0; rep <= reps; ++rep) {          // (the first launch warms up)
        CK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL(k_gather_u8, dim3(grid), dim3(256), 0, 0, idx, win, reinterpret_cast<uint8_t*>(out), accesses);
        if (mode == 1) hipLaunchKernelGGL(k_gather_u32, dim3(grid), dim3(256), 0, 0, idx, reinterpret_cast<const uint32_t*>(win), out, accesses);
        if (mode == 2) hipLaunchKernelGGL(k_scatter_u32, dim3(grid), dim3(256), 0, 0, idx, reinterpret_cast<uint32_t*>(win), accesses);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float t = 0; CK(hipEventElapsedTime(&t, a, b));
        if (rep && (best == 0 || t < best)) best = t;
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    CK(hipFree(idx)); CK(hipFree(out)); CK(hipFree(win));
    *g_per_s = (double)accesses / best / 1e6;
    return 0;
}

#ifndef PROBE_LIBRARY
int main()
{
    printf("%8s %14s %14s %14s   (G accesses / s, 2^28 per launch)\n", "W MiB", "read u8", "read u32", "write u32");
    for (uint64_t w = 16; w <= 4096; w *= 2) {
        double r[3];
        for (int m = 0; m < 3; ++m) if (msufsort_probe_random_access(0, w << 20, 1ull << 28, m, 2, &r[m])) return 1;
        printf("%8llu %14.1f %14.1f %14.1f\n", (unsigned long long)w, r[0], r[1], r[2]);
    }
    return 0;
}
#endif
