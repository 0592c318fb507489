// The write-out of the scatter levels, alone: a workgroup holds a tile of T records sorted by bin, claims a range per (tile, bin)
// from per-stripe cursors with one atomic each, and writes the tile - consecutive lanes consecutive slots, i.e. runs of T / NB
// records (on average) that abut at arbitrary 8-byte positions.  What does the store rate depend on: the run length (tile
// size), the number of bins, alignment?  (Round 4: sizing the lever "longer runs" for k_scatter0 / k_partition, profiles/HISTORY.md.)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/exp_write_runs.hip -o tools/microbench/bin/exp_write_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64; typedef unsigned u32;
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

// ALIGN: 0 = runs abut (as the kernels do), 1 = every claim rounded up to a whole 64-byte sector (8 records; wastes space, shows the cost of partial sectors)
// READ: the tile's records are read from memory too (sequentially, as k_partition reads them) instead of made up
template <int NB, int ALIGN, int READ, int NT = 0>
__global__ __launch_bounds__(1024) void k(u64* out, u32* cursors, u32 T, u32 tiles_per_stripe, u64 bin_cap, const u64* in)
{
    __shared__ u32 cnt[NB], lstart[NB + 1], gbase[NB];
    extern __shared__ u32 pad[];                             // (only there to pin the number of workgroups per CU)
    if (T == 1u) pad[threadIdx.x] = 0;
    const u32 t = threadIdx.x, tile = blockIdx.x, stripe = tile / tiles_per_stripe;
    // counts: T / NB with a jitter of +- 50 %, made to sum to T
    if (t < NB) { const u64 h = mix((u64)tile * NB + t); cnt[t] = (u32)((T / NB) / 2 + h % (T / NB + 1)); }
    __syncthreads();
    if (t == 0) { u32 s = 0; for (u32 b = 0; b < NB; ++b) { lstart[b] = s; s += cnt[b]; } lstart[NB] = s; }
    __syncthreads();
    const u32 total = lstart[NB];
    if (t < NB) {
        const u32 c = ALIGN ? ((cnt[t] + 7u) & ~7u) : cnt[t];
        gbase[t] = atomicAdd(&cursors[(u64)stripe * NB + t], c) - lstart[t];
    }
    __syncthreads();
    for (u32 s = t; s < total; s += blockDim.x) {
        u32 lo = 0, hi = NB;                       // bin of slot s
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (lstart[mid] <= s) lo = mid; else hi = mid; }
        const u64 val = READ ? in[(u64)tile * T + s] : (((u64)tile << 32) | s);
        if (NT) __builtin_nontemporal_store(val, &out[(u64)lo * bin_cap + (u32)(gbase[lo] + s)]);
        else out[(u64)lo * bin_cap + (u32)(gbase[lo] + s)] = val;
    }
}

template <int NB, int ALIGN, int READ = 0, int NT = 0> void run(u64* out, u32* cur, u32 T, u64 total_records, const u64* in = nullptr, u32 threads = 1024, u32 lds_pad = 0)
{
    const u32 stripes = 128;
    const u32 tiles = (u32)(total_records / T);
    const u32 tps = (tiles + stripes - 1) / stripes;
    const u64 bin_cap = total_records / NB * 2;                 // room per bin (all stripes of a bin share it: stripe cursors start spread out)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    static u32 h[128 * 1024];
    for (int r = 0; r < 3; ++r) {
        // stripe s of bin b starts at s * (bin_cap / stripes)
        for (u32 s = 0; s < stripes; ++s) for (u32 b = 0; b < NB; ++b) h[s * NB + b] = (u32)(s * (bin_cap / stripes));
        hipMemcpy(cur, h, sizeof(u32) * stripes * NB, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NB, ALIGN, READ, NT>), dim3(tiles), dim3(threads), lds_pad, 0, out, cur, T, tps, bin_cap, in);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    if (threads != 1024 || lds_pad) printf("[%u threads, %u KiB of LDS per workgroup] ", threads, lds_pad >> 10);
    printf("tile %6u records, %4d bins (runs of %4u records = %5u B on average), %s: %7.3f ms  %6.0f GB/s moved\n", T, NB, T / NB, T / NB * 8,
           NT ? (READ ? "read + NON-TEMPORAL stores          " : "NON-TEMPORAL stores                 ") : READ ? "records read sequentially + written" : (ALIGN ? "claims rounded to whole sectors" : "runs abut at 8-byte positions  "), best, (double)tiles * T * 8 * (READ ? 2 : 1) / best / 1e6);
}

int main()
{
    const u64 total = 1ull << 30;                               // records per launch (8 GiB written)
    u64* out; u32* cur;
    hipMalloc(&out, total * 8 * 2 + (1 << 20)); hipMalloc(&cur, sizeof(u32) * 128 * 1024);
    hipMemset(out, 0, total * 8 * 2);
    for (u32 T : {4096u, 8192u, 16384u, 32768u}) { run<256, 0>(out, cur, T, total); run<256, 1>(out, cur, T, total); }
    for (u32 T : {8192u, 16384u, 32768u}) { run<512, 0>(out, cur, T, total); }
    for (u32 T : {8192u, 16384u}) { run<128, 0>(out, cur, T, total); run<64, 0>(out, cur, T, total); }
    // with the read side (k_partition's shape): is a level that reads its records as well still faster with longer runs?
    u64* in; hipMalloc(&in, total * 8 + (1 << 20)); hipMemset(in, 1, total * 8);
    for (u32 T : {4096u, 8192u, 16384u, 32768u}) run<256, 0, 1>(out, cur, T, total, in);
    for (u32 T : {8192u, 16384u, 32768u}) run<512, 0, 1>(out, cur, T, total, in);
    run<64, 0, 1>(out, cur, 16384u, total, in);
    // workgroups per CU: 2 x 1024 threads (above) against 3 x 512 (52 KiB each), 4 x 512, 2 x 512 (80 KiB), 1 x 1024 (100 KiB)
    hipFuncSetAttribute((const void*)k<256, 0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10);
    run<256, 0>(out, cur, 16384u, total, nullptr, 512, 50 << 10);
    run<256, 0>(out, cur, 16384u, total, nullptr, 512, 0);
    run<256, 0>(out, cur, 16384u, total, nullptr, 512, 76 << 10);
    run<256, 0>(out, cur, 16384u, total, nullptr, 1024, 100 << 10);
    run<256, 0>(out, cur, 8192u, total, nullptr, 512, 50 << 10);
    run<256, 0>(out, cur, 8192u, total, nullptr, 256, 0);
    // non-temporal stores (the runs of neighbouring tiles meet in L2: does bypassing it cost or pay?)
    for (u32 T : {8192u, 16384u}) { run<256, 0, 0, 1>(out, cur, T, total); run<256, 0, 1, 1>(out, cur, T, total, in); }
    return 0;
}
