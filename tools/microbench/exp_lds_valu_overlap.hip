// Do vector ALU work and random LDS accesses of one workgroup (16 waves on a CU) overlap, or do their times add up?
// Per iteration and lane: 6 random ds_read_b32 (issued together), then NV dependent-free integer instructions per read.
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/exp_lds_valu_overlap.hip -o tools/microbench/bin/exp_lds_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NWORDS 18432
template <int NV, bool LDS>
__global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* cyc, int iters)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < NWORDS; i += 1024) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 1u, acc = 0;
    uint32_t a[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) a[b] = (x * (b + 3)) % NWORDS;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t r[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) r[b] = LDS ? lds[a[b]] : a[b] * 3u;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            uint32_t v = r[b];
#pragma unroll
            for (int q = 0; q < NV; ++q) v = (v ^ (v >> 7)) + 0x9e37u * (q + 1);      // 2 VALU per step (xor-shift fused? count below is measured)
            acc += v;
            a[b] = (v >> 3) % NWORDS;          // next address depends on the data: nothing can be hoisted
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
template <int NV, bool LDS> double run(uint32_t* out, unsigned long long* cyc)
{
    const int iters = 512, blocks = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NV, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, NWORDS * 4);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NV, LDS>), dim3(blocks), dim3(1024), NWORDS * 4, 0, out, cyc, iters);
    unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    return s / blocks / (16.0 * iters * 6);
}
int main()
{
    uint32_t* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 8);
    printf("cycles per (one LDS read + NV steps) per wave, CU-wide (16 waves):\n");
    printf("NV= 0: with LDS %6.2f   without %6.2f\n", run<0, true>(out, cyc), run<0, false>(out, cyc));
    printf("NV= 4: with LDS %6.2f   without %6.2f\n", run<4, true>(out, cyc), run<4, false>(out, cyc));
    printf("NV= 8: with LDS %6.2f   without %6.2f\n", run<8, true>(out, cyc), run<8, false>(out, cyc));
    printf("NV=16: with LDS %6.2f   without %6.2f\n", run<16, true>(out, cyc), run<16, false>(out, cyc));
    printf("NV=32: with LDS %6.2f   without %6.2f\n", run<32, true>(out, cyc), run<32, false>(out, cyc));
    return 0;
}
