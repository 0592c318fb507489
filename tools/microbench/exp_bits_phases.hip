// Phase-level model of k_sort_bits (bucket_sort_bits.hip.h): 1024 threads, 18 records per thread, 17,408 LDS words of 16
// buckets; keys come from a hash (no global memory).  Which form of phase A (mark) and phase B (rank + store) is cheapest?
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/exp_bits_phases.hip -o tools/microbench/bin/exp_bits_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
#define NW 17408
#define ITEMS 18
#define BATCH 6
__device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// AV: 0 = ds_or_rtn + late add (separate loops), 1 = ONE non-returning add {bit | 1 << 20} (guard bits 16..19), 2 = ds_or_rtn only,
//     3 = as 0 but the late add inside the issue loop
template <int AV, int BV>
__global__ __launch_bounds__(1024) void k(u32* outg, unsigned long long* cyc, int segs)
{
    extern __shared__ u32 lds[];
    u32* bw = lds; u32* out = lds + NW + 4;
    const u32 t = threadIdx.x, lane = t & 63;
    unsigned long long ta = 0, tb = 0, tc = 0;
    u32 acc = 0;
    for (int s = 0; s < segs; ++s) {
        u32 key[ITEMS], idx[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { const u32 h = hash((blockIdx.x * segs + s) * 18432u + j * 1024u + t); key[j] = (h >> 8) ; idx[j] = h; }
        for (u32 i = t; i < (NW + 4) / 4; i += 1024) reinterpret_cast<uint4*>(bw)[i] = uint4{0, 0, 0, 0};
        __syncthreads();
        unsigned long long t0 = clock64();
        const u32 sh = 10;
#pragma unroll
        for (int j0 = 0; j0 < ITEMS; j0 += BATCH) {
            u32 old[BATCH], bit[BATCH], wa[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int j = j0 + b;
                const u32 kk = ((j >> 1) * 2048u + 2 * t + (j & 1)) < 16384u ? __umul24(key[j] & 0xffffffu, 17u) : ((NW + 1) * 16u) << sh;
                key[j] = kk;
                const u32 g = kk >> sh;
                bit[b] = 1u << (g & 15u); wa[b] = g >> 4;
                if (AV == 1) atomicAdd(&bw[wa[b]], bit[b] | 0x100000u);
                else old[b] = atomicOr(&bw[wa[b]], bit[b]);
                if (AV == 3) { if (old[b] & bit[b]) atomicAdd(&bw[wa[b]], 0x10000u); }
            }
            if (AV == 0) {
#pragma unroll
                for (int b = 0; b < BATCH; ++b) if (old[b] & bit[b]) atomicAdd(&bw[wa[b]], 0x10000u);
            }
            if (AV == 2) {
#pragma unroll
                for (int b = 0; b < BATCH; ++b) acc += old[b];
            }
        }
        __syncthreads();
        unsigned long long t1 = clock64();
        // scan stand-in: mark every word clean with first row = 0.9 * word index (what matters is phase B's access pattern)
        for (u32 i = t; i < NW + 2; i += 1024) { const u32 w = bw[i]; bw[i] = (w & 0xffffu) | (((i * 15u) >> 4) << 16) | ((AV != 1 && (w >> 16)) || (AV == 1 && __popc(w & 0xfffffu) != (w >> 20)) ? 0x80000000u : 0u); }
        __syncthreads();
        unsigned long long t2 = clock64();
        u32 wtot = 0;
        unsigned long long dball[ITEMS];
#pragma unroll
        for (int j0 = 0; j0 < ITEMS; j0 += BATCH) {
            u32 e[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) e[b] = bw[(key[j0 + b] >> sh) >> 4];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int j = j0 + b;
                const u32 g = key[j] >> sh;
                const u32 row = (u32)__popc(e[b] & ((1u << (g & 15u)) - 1u)) + ((e[b] >> 16) & 0x7fffu);
                const bool dirty = (int)e[b] < 0;
                if (BV == 0) { if (!dirty) out[row] = idx[j]; dball[j] = __ballot(dirty); wtot += (u32)__popcll(dball[j]); }
                if (BV == 1) out[row] = idx[j];
                if (BV == 2) acc += row;
            }
        }
        if (BV == 0 && wtot) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&out[17500], wtot);
            u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)base) & 1023u;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (dball[j]) {
                    const unsigned long long bal = dball[j];
                    const u32 pos = wbase + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                    if ((bal >> lane) & 1ull) { uint2 r; r.x = key[j]; r.y = idx[j]; reinterpret_cast<uint2*>(out + 17600)[pos] = r; }
                    wbase += (u32)__popcll(bal);
                }
        }
        __syncthreads();
        unsigned long long t3 = clock64();
        ta += t1 - t0; tb += t3 - t2; tc += t2 - t1;
        acc += out[t] + bw[t];
    }
    if (t == 0) { cyc[blockIdx.x * 3] = ta; cyc[blockIdx.x * 3 + 1] = tb; cyc[blockIdx.x * 3 + 2] = tc; }
    outg[blockIdx.x * 1024 + t] = acc;
}
template <int AV, int BV> void run(const char* name, u32* out, unsigned long long* cyc)
{
    const int segs = 64, blocks = 256;
    const size_t lds = (NW + 4 + 17408 + 64 + 2048 * 2 + 1024) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<AV, BV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<AV, BV>), dim3(blocks), dim3(1024), lds, 0, out, cyc, segs);
    unsigned long long h[256 * 3];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0; for (int i = 0; i < blocks; ++i) { a += (double)h[3 * i]; b += (double)h[3 * i + 1]; c += (double)h[3 * i + 2]; }
    printf("%-64s A %7.0f  B %7.0f  (stand-in scan %6.0f) cycles per segment\n", name, a / blocks / segs, b / blocks / segs, c / blocks / segs);
}
int main()
{
    u32* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 3 * 8);
    run<0, 0>("A: or_rtn, late adds after the batch | B: store + ballots + push", out, cyc);
    run<3, 1>("A: or_rtn, late add inside the loop  | B: store only", out, cyc);
    run<2, 2>("A: or_rtn only                       | B: rank only (no store)", out, cyc);
    run<1, 0>("A: one non-returning add (guard bits)| B: store + ballots + push", out, cyc);
    run<1, 1>("A: one non-returning add (guard bits)| B: store only", out, cyc);
    return 0;
}
