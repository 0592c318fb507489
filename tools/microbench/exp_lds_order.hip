// Are returning LDS atomics of ONE wave instruction served in lane order when several lanes hit the same address?
// (An LSD radix pass wants stable ranks; if the hardware serialises same-address lanes lowest-lane-first, the value an
// atomicAdd returns IS the stable rank and the 8 ballots per element of the match-any scheme are not needed.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/exp_lds_order.hip -o tools/microbench/bin/exp_lds_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* dig, uint32_t* bad, int iters, int nbins)
{
    __shared__ uint32_t cnt[4][256];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t mism = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = lane; i < 256; i += 64) cnt[w][i] = 0;
        __syncthreads();
        uint32_t seen[8];
        for (int j = 0; j < 8; ++j) {
            const uint32_t d = dig[((size_t)(blockIdx.x * iters + it) * 8 + j) * 256 + threadIdx.x] % nbins;
            const uint32_t old = atomicAdd(&cnt[w][d], 1u);
            // expected stable rank: elements of earlier instructions with this digit + lower lanes of this one
            uint64_t m = ~0ull;
            for (int b = 0; b < 8; ++b) { const bool bit = (d >> b) & 1; const uint64_t bal = __ballot(bit); m &= bit ? bal : ~bal; }
            const uint32_t below = __popcll(m & ((1ull << lane) - 1));
            seen[j] = old - below;              // must equal the count before this instruction = same for all lanes of the digit
            const uint32_t first = __shfl(seen[j], __ffsll((long long)m) - 1, 64);
            // count before this instruction: track with a second, ballot-based counter
            if (seen[j] != first) ++mism;
            // and lanes sharing a digit must get consecutive values starting at `first`
            if (old != first + below) ++mism;
        }
        __syncthreads();
    }
    if (mism) atomicAdd(bad, mism);
}
int main()
{
    const int blocks = 1024, iters = 64;
    size_t n = (size_t)blocks * iters * 8 * 256;
    uint32_t* h = (uint32_t*)malloc(n * 4);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 33); }
    uint32_t *d, *bad;
    hipMalloc(&d, n * 4); hipMalloc(&bad, 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    for (int nbins : {1, 2, 3, 7, 16, 64, 256}) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, bad, iters, nbins);
        uint32_t r = 0;
        hipMemcpy(&r, bad, 4, hipMemcpyDeviceToHost);
        printf("bins %3d: %u mismatches in %zu returning LDS atomics\n", nbins, r, n);
    }
    return 0;
}
