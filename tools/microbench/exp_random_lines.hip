// Rate of random 4-byte reads (one 128-byte line each) against the size of the region they fall into: L2 (4 MiB per XCD),
// the memory-side Infinity Cache (256 MiB) and HBM.  Every lane reads K independent addresses per iteration (K loads in
// flight per lane), 256 CUs x 8 workgroups x 256 threads.  Second table: the same reads where each lane re-reads, DELAY
// iterations later, a line it has touched before (what a "fetch the preceding characters when the row becomes final" step
// would see: the line came in when the key was gathered).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/exp_random_lines.hip -o tools/microbench/bin/exp_random_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#ifndef K
#define K 8
#endif
__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ __launch_bounds__(256) void k_rand(const uint32_t* __restrict__ buf, uint64_t lines, int iters, uint32_t* out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t r[K];
#pragma unroll
        for (int b = 0; b < K; ++b) {
            const uint64_t h = mix(gid * 1315423911ull + (uint64_t)(it * K + b) * 0x9E3779B97F4A7C15ull);
            r[b] = buf[(h % lines) * 32 + (h >> 59)];
        }
#pragma unroll
        for (int b = 0; b < K; ++b) acc += r[b];
    }
    out[gid] = acc;
}
// every iteration: K fresh random lines over the whole region + K re-reads of the lines of `delay` iterations ago
__global__ __launch_bounds__(256) void k_reread(const uint32_t* __restrict__ buf, uint64_t lines, int iters, int delay, uint32_t* out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t r[2 * K];
#pragma unroll
        for (int b = 0; b < K; ++b) {
            const uint64_t h = mix(gid * 1315423911ull + (uint64_t)(it * K + b) * 0x9E3779B97F4A7C15ull);
            r[b] = buf[(h % lines) * 32 + (h >> 59)];
            const uint64_t g = mix(gid * 1315423911ull + (uint64_t)((it - delay) * K + b) * 0x9E3779B97F4A7C15ull);
            r[K + b] = it >= delay ? buf[(g % lines) * 32 + ((g >> 59) ^ 1)] : 0u;
        }
#pragma unroll
        for (int b = 0; b < 2 * K; ++b) acc += r[b];
    }
    out[gid] = acc;
}
// K dependent chains per lane: the address of a chain's next read comes from the value it has just read (the inverse BWT walk)
__global__ __launch_bounds__(256) void k_chain(const uint32_t* __restrict__ buf, uint64_t lines, int iters, uint32_t* out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t r[K];
#pragma unroll
    for (int b = 0; b < K; ++b) r[b] = (uint32_t)b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < K; ++b) {
            const uint64_t h = mix(gid * 1315423911ull + (uint64_t)(it * K + b) * 0x9E3779B97F4A7C15ull + r[b]);
            r[b] = buf[(h % lines) * 32 + (h >> 59)];
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int b = 0; b < K; ++b) acc += r[b];
    out[gid] = acc;
}
int main()
{
    const size_t maxb = 8ull << 30;
    uint32_t *buf, *out;
    hipMalloc(&buf, maxb); hipMemset(buf, 1, maxb);
    const int blocks = 256 * 8;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("random 4-byte reads, one 128-byte line each, %d loads in flight per lane, %d workgroups of 256\n", K, blocks);
    for (size_t mb : {1, 4, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 4096, 8192}) {
        const uint64_t lines = (mb << 20) / 128;
        const int iters = 256;
        hipLaunchKernelGGL(k_rand, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rand, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double acc = (double)blocks * 256 * iters * K;
        printf("region %5zu MiB: %7.2f G reads/s  (%6.2f TB/s of lines)\n", mb, acc / ms / 1e6, acc * 128 / ms / 1e9);
    }
    printf("\n%d dependent chains per lane (next address from the value read), 2048 lanes per CU\n", K);
    for (size_t mb : {1024, 4096}) {
        const uint64_t lines = (mb << 20) / 128;
        const int iters = 256;
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double acc = (double)blocks * 256 * iters * K;
        printf("region %5zu MiB: %7.2f G hops/s\n", mb, acc / ms / 1e6);
    }
    printf("\nfresh random lines over 1 GiB + a re-read of the lines of `delay` iterations ago (other word of the line)\n");
    for (int delay : {0, 1, 4, 16, 64}) {
        const uint64_t lines = (1024ull << 20) / 128;
        const int iters = 256;
        hipLaunchKernelGGL(k_reread, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, delay, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_reread, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, delay, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double acc = (double)blocks * 256 * iters * K;
        // chip-wide lines touched between a read and its re-read: delay x (blocks x 256 x K)
        printf("delay %3d iterations (%8.1f MiB of lines in between): %7.2f G fresh reads/s (each with its re-read)\n", delay,
               (double)delay * blocks * 256 * K * 128 / 1048576.0, acc / ms / 1e6);
    }
    return 0;
}
