// What bounds k_hist16 on uniform random bytes?  (round-5 review item 8: "k_hist16 back above 0.45, or a micro-benchmark that measures the
// 65,536-bin LDS-atomic ceiling at this occupancy and shows the kernel is on it".)
// Same shape as the kernel: one 1024-thread workgroup per CU (128 KiB of LDS counters), 16 adds per lane and step, no global loads in A - D:
//   A  ds_add_u32, non-returning, random words of the 32,768 (what k_hist16 issues; 64 random lanes on 32 banks collide)
//   B  the same with conflict-free addresses (lane l of a half-wave on bank l): the LDS pipe without bank conflicts
//   C  ds_add_u64 on 16,384 random 8-byte words (four 16-bit counters per word): would wider words dodge the conflicts?
//   D  A plus the kernel's own arithmetic per key (h16_key / h16_add on a 16-byte register vector + look-ahead word)
//   E  k_hist16<0> itself on 2^30 - 1 random bytes in HBM
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/exp_lds_hist_ceiling.hip -o tools/microbench/bin/exp_lds_hist_ceiling
#include "../../msufsort_amd/csrc/sa_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32 xs(u32& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int MODE>
__global__ __launch_bounds__(1024) void k_ceiling(u32 steps, u32* sink)
{
    extern __shared__ u32 lds[];
    const u32 t = threadIdx.x;
    for (u32 i = t; i < 32768u; i += 1024u) lds[i] = 0;
    __syncthreads();
    u32 s = (blockIdx.x * 1024u + t) * 2654435761u + 12345u;
    for (u32 it = 0; it < steps; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { const u32 r = xs(s); atomicAdd(&lds[r & 0x7fffu], 1u); }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { const u32 r = xs(s); atomicAdd(&lds[((r & 0x3ffu) << 5) | (t & 31u)], 1u); }
        } else if (MODE == 2) {
            unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
#pragma unroll
            for (int j = 0; j < 16; ++j) { const u32 r = xs(s); atomicAdd(&l64[r & 0x3fffu], 1ull << (16u * ((r >> 14) & 3u))); }
        } else {
            const u32 w[6] = {xs(s), xs(s), xs(s), xs(s), xs(s), 0u};
#pragma unroll
            for (int j = 0; j < 16; ++j) h16_add(lds, h16_key(w, j), 1u);
        }
    }
    __syncthreads();
    u32 a = 0;
    for (u32 i = t; i < 32768u; i += 1024u) a += lds[i];
    if (a == 0xdeadbeefu) sink[0] = a;
}

template <int MODE>
static double run(const char* name, u32 steps, u32* sink)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ceiling<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 8256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        k_ceiling<MODE><<<256, 1024, 131072 + 8256>>>(steps, sink);      // (+ the kernel's overflow list: the same one workgroup per CU)
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double keys = 256.0 * 1024 * 16 * steps;
    printf("%-62s %8.3f ms  %7.2f T adds/s (= TB/s of text at one key per byte; %.3f of the 8 TB/s HBM peak)\n", name, best, keys / best / 1e9, keys / best / 1e9 / 8.0);
    return keys / best / 1e9;
}

int main()
{
    u32* sink; CK(hipMalloc(&sink, 4));
    const u32 steps = 256;          // 256 x 16 KiB = 4 MiB of keys per workgroup: one chunk of a 1 GiB text
    run<0>("A  ds_add_u32, random words (bank conflicts as in the kernel)", steps, sink);
    run<1>("B  ds_add_u32, conflict-free addresses", steps, sink);
    run<2>("C  ds_add_u64, random 8-byte words (4 x 16-bit counters)", steps, sink);
    run<3>("D  A + the kernel's key arithmetic (h16_key / h16_add)", steps, sink);
    // E: the kernel
    const u64 n = (1ull << 30) - 1;
    std::vector<u8> h(n + 64, 0);
    u64 s = 12345;
    for (u64 i = 0; i < n; i += 8) { u64 z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; memcpy(&h[i], &z, std::min<u64>(8, n - i)); }
    u8* d; CK(hipMalloc(&d, n + 64)); CK(hipMemcpy(d, h.data(), n + 64, hipMemcpyHostToDevice));
    u32* p1; CK(hipMalloc(&p1, 256ull * 65536 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist16<0>), hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS_BYTES));
    const u32 m = (u32)n;
    u32 nc = 128; u64 cl = ((u64)m + nc - 1) / nc; cl = (cl + 32767) / 32768 * 32768;
    const u32 per = 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        k_hist16<0><<<nc * per, 1024, H16_LDS_BYTES>>>(d, m, (u32)(cl / per), nc * per, p1, 0u, (const unsigned short*)nullptr, 0u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-62s %8.3f ms  %7.2f TB/s of text (%.3f of the HBM peak)\n", "E  k_hist16<0>, 2^30 - 1 random bytes from HBM", best, n / best / 1e9, n / best / 1e9 / 8.0);
    return 0;
}
