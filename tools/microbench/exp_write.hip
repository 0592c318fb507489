// Write-bandwidth probes: streaming fill with 16-B and 8-B stores, and 256-B runs at scattered places.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64; typedef unsigned u32;
__global__ __launch_bounds__(1024) void fill16(uint4* p, u64 n16) { for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (u64)gridDim.x * 1024) p[i] = make_uint4((u32)i, 1, 2, 3); }
__global__ __launch_bounds__(1024) void fill8(u64* p, u64 n8) { for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n8; i += (u64)gridDim.x * 1024) p[i] = i; }
// each wave writes RUN-byte runs (8 B per lane) at pseudo-random run-aligned places: the scatter kernels' store shape
template <int RUN> __global__ __launch_bounds__(1024) void runs8(u64* p, u64 n8, u32 shift)
{
    const u64 nruns = n8 / (RUN / 8);
    const u32 lanes_per_run = RUN / 8;
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n8; i += (u64)gridDim.x * 1024) {
        const u64 run = i / lanes_per_run, l = i % lanes_per_run;
        const u64 dst = (run * 0x9E3779B97F4A7C15ull >> shift) % nruns;     // a permutation-like hash of the run number
        p[dst * lanes_per_run + l] = i;
    }
}
__global__ __launch_bounds__(1024) void copy16(const uint4* s, uint4* d, u64 n16) { for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (u64)gridDim.x * 1024) d[i] = s[i]; }
__global__ __launch_bounds__(1024) void read16(const uint4* s, u64 n16, u32* sink) { u32 a = 0; for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (u64)gridDim.x * 1024) { uint4 v = s[i]; a += v.x ^ v.y ^ v.z ^ v.w; } if (a == 0x12345) *sink = a; }
int main()
{
    const u64 bytes = 8ull << 30;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); u32* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double gb, auto f) { float best = 1e9; for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } printf("%-28s %8.3f ms  %7.0f GB/s\n", name, best, gb / best * 1e3); };
    const double GB = bytes / 1e9;
    for (int grid : {2048, 16384}) {
        printf("grid %d\n", grid);
        time("fill 16 B/lane", GB, [&] { fill16<<<grid, 1024>>>((uint4*)a, bytes / 16); });
        time("fill 8 B/lane", GB, [&] { fill8<<<grid, 1024>>>((u64*)a, bytes / 8); });
        time("runs 512 B (8 B/lane)", GB, [&] { runs8<512><<<grid, 1024>>>((u64*)a, bytes / 8, 20); });
        time("runs 256 B (8 B/lane)", GB, [&] { runs8<256><<<grid, 1024>>>((u64*)a, bytes / 8, 20); });
        time("runs 128 B (8 B/lane)", GB, [&] { runs8<128><<<grid, 1024>>>((u64*)a, bytes / 8, 20); });
        time("runs 64 B (8 B/lane)", GB, [&] { runs8<64><<<grid, 1024>>>((u64*)a, bytes / 8, 20); });
        time("read 16 B/lane", GB, [&] { read16<<<grid, 1024>>>((const uint4*)a, bytes / 16, sink); });
        time("copy 16 B/lane (r+w bytes)", 2 * GB, [&] { copy16<<<grid, 1024>>>((const uint4*)a, (uint4*)b, bytes / 16); });
    }
    return 0;
}
