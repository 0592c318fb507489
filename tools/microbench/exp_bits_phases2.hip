// Phase-level model of k_sort_bits, second edition: full compute path on hashed keys (no global memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/exp_bits_phases2.hip -o tools/microbench/bin/exp_bits_phases2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef unsigned long long u64;
#define THREADS 1024
#define WPT 17
#define NW (THREADS * WPT)
#define ITEMS 18
#define BATCH 6
#define LCAP 2048
#define LEN_MAX 17408
__device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
// PV (push variant): 0 = 64-bit ballots kept per item + mbcnt; 1 = per-thread mask, wave scan of the per-thread counts
template <int PV>
__global__ __launch_bounds__(THREADS) void k(u32* outg, u64* cyc, int segs, u32* chk)
{
    extern __shared__ u32 lds[];
    u32* bw = lds; u32* out = lds + NW + 4; uint2* lst = reinterpret_cast<uint2*>(out + LEN_MAX + 64); u32* tot = reinterpret_cast<u32*>(lst + LCAP); u32* misc = tot + 16;
    const u32 t = threadIdx.x, lane = t & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    u64 tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 acc = 0, dirty_total = 0;
    const u32 len = 16384, sh = 10;
    for (int s = 0; s < segs; ++s) {
        u32 key[ITEMS], idx[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { const u32 h = hash((blockIdx.x * segs + s) * 18432u + j * 1024u + t); key[j] = (h >> 8); idx[j] = ((j >> 1) * 2048u + 2 * t + (j & 1)); }
        u64 c0 = clock64();
        if (t < 16) misc[t] = 0;
        for (u32 i = t; i < (NW + 4) / 4; i += THREADS) reinterpret_cast<uint4*>(bw)[i] = uint4{0, 0, 0, 0};
        __syncthreads();
        u64 c1 = clock64(); tp[0] += c1 - c0;
        // ---- A: ONE non-returning add per record: bit | 1 << 20 (bits 16..19 catch the carries of colliding adds)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 p = ((j >> 1) * 2048u + 2 * t + (j & 1));
            const u32 kk = p < len ? __umul24(key[j] & 0xffffffu, (u32)WPT) : ((NW + 1) * 16u) << sh;
            key[j] = kk;
            const u32 g = kk >> sh;
            atomicAdd(&bw[g >> 4], (1u << (g & 15u)) | 0x100000u);
        }
        __syncthreads();
        u64 c2 = clock64(); tp[1] += c2 - c1;
        // ---- S
        {
            u32 sum = 0;
#pragma unroll
            for (int k2 = 0; k2 < WPT; ++k2) sum += bw[t * WPT + k2] >> 20;
            const u32 inc = wave_incl_scan_dpp(sum);
            if (lane == 63) tot[wv] = inc;
            __syncthreads();
            u32 x = inc - sum;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) if (k2 < (int)wv) x += tot[k2];
#pragma unroll
            for (int k2 = 0; k2 < WPT; ++k2) {
                const u32 e = bw[t * WPT + k2];
                const u32 cnt = e >> 20;
                const bool dirty = (u32)__popc(e & 0xfffffu) != cnt;
                bw[t * WPT + k2] = (e & 0xffffu) | ((x | (dirty ? 0x8000u : 0u)) << 16);
                x += cnt;
            }
            if (t == THREADS - 1) { bw[NW] = x << 16; bw[NW + 1] = (LEN_MAX + 32u) << 16; }
            __syncthreads();
        }
        u64 c3 = clock64(); tp[2] += c3 - c2;
        // ---- B
        if (PV == 0) {
            u64 dball[ITEMS]; u32 wtot = 0;
#pragma unroll
            for (int j0 = 0; j0 < ITEMS; j0 += BATCH) {
                u32 e[BATCH];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) e[b] = bw[(key[j0 + b] >> sh) >> 4];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int j = j0 + b; const u32 g = key[j] >> sh;
                    const u32 row = (u32)__popc(e[b] & ((1u << (g & 15u)) - 1u)) + ((e[b] >> 16) & 0x7fffu);
                    const bool dirty = (int)e[b] < 0;
                    if (!dirty) out[row] = idx[j];
                    dball[j] = __ballot(dirty); wtot += (u32)__popcll(dball[j]);
                }
            }
            if (wtot) {
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&misc[0], wtot);
                u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)base);
                if (wbase + wtot <= LCAP) {
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j)
                        if (dball[j]) {
                            const u64 bal = dball[j];
                            const u32 pos = wbase + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                            if ((bal >> lane) & 1ull) { uint2 r; r.x = key[j]; r.y = idx[j]; lst[pos] = r; }
                            wbase += (u32)__popcll(bal);
                        }
                } else misc[1] = 1;
            }
        } else {
            u32 dmask = 0;
#pragma unroll
            for (int j0 = 0; j0 < ITEMS; j0 += BATCH) {
                u32 e[BATCH];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) e[b] = bw[(key[j0 + b] >> sh) >> 4];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int j = j0 + b; const u32 g = key[j] >> sh;
                    const u32 row = (u32)__popc(e[b] & ((1u << (g & 15u)) - 1u)) + ((e[b] >> 16) & 0x7fffu);
                    if ((int)e[b] >= 0) out[row] = idx[j];
                    dmask |= (e[b] >> 31) << j;
                }
            }
            const u32 dc = (u32)__popc(dmask);
            const u32 inc = wave_incl_scan_dpp(dc);
            const u32 wtot = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            if (wtot) {
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&misc[0], wtot);
                const u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)base);
                if (wbase + wtot <= LCAP) {
                    u32 pos = wbase + inc - dc;
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j)
                        if ((dmask >> j) & 1u) { uint2 r; r.x = key[j]; r.y = idx[j]; lst[pos] = r; ++pos; }
                } else misc[1] = 1;
            }
        }
        __syncthreads();
        u64 c4 = clock64(); tp[3] += c4 - c3;
        const u32 nl = misc[0];
        dirty_total += nl;
        // ---- D
        {
            u32 dk[2], di[2], dw[2], db[2], dc2[2], ds[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32 e = t + (u32)i * THREADS;
                dk[i] = 0; di[i] = 0; dw[i] = 0; db[i] = 0; dc2[i] = 0; ds[i] = 0;
                if (e < nl) {
                    const uint2 r = lst[e];
                    dk[i] = r.x; di[i] = r.y; dw[i] = (r.x >> sh) >> 4;
                    const u32 w0 = bw[dw[i]], w1 = bw[dw[i] + 1u];
                    db[i] = (w0 >> 16) & 0x7fffu; dc2[i] = ((w1 >> 16) & 0x7fffu) - db[i];
                    bw[dw[i]] = w0 & 0xffff0000u;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (t + (u32)i * THREADS < nl) { ds[i] = atomicAdd(&bw[dw[i]], 1u) & 0xffffu; out[db[i] + ds[i]] = dk[i]; }
            __syncthreads();
            u32 dl[2], dq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                dl[i] = 0; dq[i] = 0;
                const bool in = t + (u32)i * THREADS < nl;
                u32 c[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) c[q] = out[db[i] + q];
                u32 lt = 0, eq = 0, ro = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { const bool v = (u32)q < dc2[i]; lt += v & (c[q] < dk[i]); eq += v & (c[q] == dk[i]); ro += v & (c[q] == dk[i]) & ((u32)q < ds[i]); }
                if (__ballot(in && dc2[i] > 8u)) {
#pragma nounroll
                    for (u32 q = 8; q < dc2[i]; ++q) { const u32 cc = out[db[i] + q]; lt += cc < dk[i]; eq += cc == dk[i]; ro += (cc == dk[i]) & (q < ds[i]); }
                }
                if (in) { dl[i] = lt; dq[i] = eq | (ro << 16); }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (t + (u32)i * THREADS < nl) { const u32 ro = dq[i] >> 16; out[db[i] + dl[i] + ro] = di[i]; if ((dq[i] & 0xffffu) > 1u) atomicAdd(&misc[4], 1u); }
            __syncthreads();
        }
        u64 c5 = clock64(); tp[4] += c5 - c4;
        // ---- out (LDS side only) + a check that the rows are a sorted permutation: out[r] = source position -> its key must be non-decreasing
#pragma unroll
        for (int q = 0; q < ITEMS / 2; ++q) { const u32 p = (q * 2048u + 2 * t); if (p + 1 < len) { acc += out[p] ^ out[p + 1]; } }
        __syncthreads();
        u64 c6 = clock64(); tp[5] += c6 - c5;
    }
    if (t == 0) { for (int i = 0; i < 6; ++i) cyc[blockIdx.x * 8 + i] = tp[i]; cyc[blockIdx.x * 8 + 6] = dirty_total; }
    outg[blockIdx.x * THREADS + t] = acc;
}
template <int PV> void run(const char* name, u32* out, u64* cyc)
{
    const int segs = 64, blocks = 256;
    const size_t lds = (NW + 4 + LEN_MAX + 64 + LCAP * 2 + 64) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<PV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<PV>), dim3(blocks), dim3(THREADS), lds, 0, out, cyc, segs, (u32*)nullptr);
    u64 h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double a[8] = {0}; for (int i = 0; i < blocks; ++i) for (int q = 0; q < 7; ++q) a[q] += (double)h[8 * i + q];
    printf("%-40s clear %5.0f  A %5.0f  scan %5.0f  B %5.0f  D %5.0f  out %5.0f  = %6.0f cycles per segment; dirty records per segment %.0f\n", name,
           a[0] / blocks / segs, a[1] / blocks / segs, a[2] / blocks / segs, a[3] / blocks / segs, a[4] / blocks / segs, a[5] / blocks / segs,
           (a[0] + a[1] + a[2] + a[3] + a[4] + a[5]) / blocks / segs, a[6] / blocks / segs);
}
int main()
{
    u32* out; u64* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("push: ballots per item + mbcnt", out, cyc);
    run<1>("push: per-thread mask + wave scan", out, cyc);
    return 0;
}
