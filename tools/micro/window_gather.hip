// Micro-benchmark: random byte reads / 4-byte writes confined to a window of W MiB (is the memory-side cache a lever?).
// hipcc --offload-arch=gfx950 -O3 tools/micro/window_gather.hip -o gpurun_out/window_gather && gpurun_out/window_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
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
int main()
{
    const uint64_t n = 1ull << 28;
    uint32_t *idx, *out32; uint8_t *src, *out8;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out32, n * 4)); CK(hipMalloc(&out8, n)); CK(hipMalloc(&src, 4ull << 30));
    CK(hipMemset(src, 1, 4ull << 30));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    printf("%8s %14s %14s %14s   (G accesses / s, n = 2^28 per launch)\n", "W MiB", "read u8", "read u32", "write u32");
    for (uint64_t w = 16; w <= 4096; w *= 2) {
        hipLaunchKernelGGL(k_fill_idx, dim3(grid), dim3(256), 0, 0, idx, n, (w << 20) - 1, w);
        float t[3];
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
                if (k == 0) hipLaunchKernelGGL(k_gather_u8, dim3(grid), dim3(256), 0, 0, idx, src, out8, n);
                if (k == 1) hipLaunchKernelGGL(k_gather_u32, dim3(grid), dim3(256), 0, 0, idx, (const uint32_t*)src, out32, n);
                if (k == 2) hipLaunchKernelGGL(k_scatter_u32, dim3(grid), dim3(256), 0, 0, idx, (uint32_t*)src, n);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&t[k], a, b));
            }
        }
        printf("%8llu %14.1f %14.1f %14.1f\n", (unsigned long long)w, n / t[0] / 1e6, n / t[1] / 1e6, n / t[2] / 1e6);
    }
    return 0;
}
