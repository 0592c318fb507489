// The store shape of the scatter levels: many output streams (cursor = atomic counter), a wave's half (32 lanes) claims a run of
// `len` 8-byte records from a pseudo-random stream and writes it.  MODE 0: len = 32 (every run 256 bytes, sector aligned);
// MODE 1: len = 17 .. 47 (runs abut at arbitrary 8-byte positions: first and last 64-byte sector of a run are shared with the
// neighbouring claims, which other workgroups write at other times); MODE 2: the same lengths rounded to multiples of 8 records
// (abutting runs, every claim a whole number of sectors).  What does the partial-sector sharing cost?
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/exp_write_streams.hip -o tools/microbench/bin/exp_write_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64; typedef unsigned u32;
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
template <int MODE>
__global__ __launch_bounds__(1024) void k(u64* out, u32* cursors, u32 nstreams, u32 stream_cap, int iters, u32 local)
{
    const u32 half = (blockIdx.x * 1024u + threadIdx.x) >> 5, l = threadIdx.x & 31u;
    for (int it = 0; it < iters; ++it) {
        const u64 h = mix((u64)half * 0x9E3779B97F4A7C15ull + (u64)it * 1315423911ull);
        // local != 0: the workgroups of one XCD (blockIdx & 7) keep to their own eighth of the streams, as xcd_tile arranges it
        u32 s = (u32)(h % nstreams);
        if (local) s = (s & ~7u) | (blockIdx.x & 7u);
        u32 len = 32;
        if (MODE == 1) len = 17u + (u32)((h >> 40) % 31u);
        if (MODE == 2) len = 16u + 8u * (u32)((h >> 40) % 4u);
        u32 base = 0;
        if (l == 0) base = atomicAdd(&cursors[s], len);
        base = __shfl(base, 0, 32);
        if (base + len > stream_cap) continue;
        u64* dst = out + (u64)s * stream_cap + base;
        if (l < len) dst[l] = h + l;
        if (l + 32 < len) dst[l + 32] = h + l + 32;
    }
}
template <int MODE> void run(const char* name, u64* out, u32* cur, u32 nstreams, u32 cap, u32 local)
{
    const int blocks = 2048, iters = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipMemset(cur, 0, (size_t)nstreams * 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, cur, nstreams, cap, iters, local);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // bytes written = sum of the cursors x 8
    static u32 h[1 << 16];
    hipMemcpy(h, cur, (size_t)nstreams * 4, hipMemcpyDeviceToHost);
    double recs = 0; for (u32 i = 0; i < nstreams; ++i) recs += h[i] < cap ? h[i] : cap;
    printf("%-58s %7.3f ms  %6.0f GB/s written\n", name, best, recs * 8 / best / 1e6);
}
int main()
{
    const u32 nstreams = 32768, cap = 1u << 15;              // 32768 streams x 256 KiB = 8 GiB
    u64* out; u32* cur;
    hipMalloc(&out, (size_t)nstreams * cap * 8); hipMalloc(&cur, (size_t)nstreams * 4);
    hipMemset(out, 0, (size_t)nstreams * cap * 8);
    for (u32 local : {0u, 1u}) {
        printf("%s\n", local ? "streams private to an XCD (blockIdx & 7)" : "any workgroup writes any stream");
        run<0>("runs of 32 records = 256 B, sector aligned", out, cur, nstreams, cap, local);
        run<1>("runs of 17 .. 47 records, abutting at 8-byte positions", out, cur, nstreams, cap, local);
        run<2>("runs of 16 / 24 / 32 / 40 records, abutting, whole sectors", out, cur, nstreams, cap, local);
    }
    return 0;
}
