// LDS throughput of the access shapes the bucket sort chooses between: cycles per wave-instruction with 16 waves (one
// 1024-thread workgroup) per CU, random word addresses in a 72 KiB table, K operations per lane issued in batches of 6.
//   0 ds_or_rtn_b32   1 ds_add_u32 (no return)   2 ds_read_b32   3 ds_read_b64   4 ds_write_b32   5 ds_write_b64
//   6 ds_add_rtn_u32  7 ds_or_b32 (no return)    8 ds_read_u8    9 ds_write_b8   10 ds_read_b32 linear   11 ds_max_u32 (no return)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/exp_lds_rates.hip -o tools/microbench/bin/exp_lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NWORDS 18432
template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* cyc, int iters)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < NWORDS; i += 1024) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 1u, acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t a[6], r[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            a[b] = OP == 10 ? ((threadIdx.x + 64u * (uint32_t)(it * 6 + b)) % NWORDS) : (x % NWORDS);
            if (OP == 3 || OP == 5) a[b] &= ~1u;
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            r[b] = 0;
            if (OP == 0) r[b] = atomicOr(&lds[a[b]], 1u << (x & 31));
            else if (OP == 1) atomicAdd(&lds[a[b]], 1u);
            else if (OP == 2 || OP == 10) r[b] = lds[a[b]];
            else if (OP == 3) { const uint2 v = *reinterpret_cast<const uint2*>(&lds[a[b]]); r[b] = v.x ^ v.y; }
            else if (OP == 4) lds[a[b]] = x;
            else if (OP == 5) { uint2 v; v.x = x; v.y = a[b]; *reinterpret_cast<uint2*>(&lds[a[b]]) = v; }
            else if (OP == 6) r[b] = atomicAdd(&lds[a[b]], 1u);
            else if (OP == 7) atomicOr(&lds[a[b]], 1u << (x & 31));
            else if (OP == 8) r[b] = reinterpret_cast<const unsigned char*>(lds)[a[b] * 4 + (x & 3)];
            else if (OP == 9) reinterpret_cast<unsigned char*>(lds)[a[b] * 4 + (x & 3)] = (unsigned char)x;
            else if (OP == 11) atomicMax(&lds[a[b]], x);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) acc += r[b];
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = acc + lds[threadIdx.x];
}
template <int OP> void run(const char* name, uint32_t* out, unsigned long long* cyc)
{
    const int iters = 512, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, NWORDS * 4);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), NWORDS * 4, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), NWORDS * 4, 0, out, cyc, iters);
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    // 16 waves x iters x 6 wave-instructions per workgroup
    printf("%-28s %7.2f cycles per wave-instruction (CU-wide, incl. ~8 VALU of address generation per op)\n", name, s / blocks / (16.0 * iters * 6));
}
int main()
{
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("ds_or_rtn_b32 random", out, cyc);
    run<7>("ds_or_b32 random (no rtn)", out, cyc);
    run<6>("ds_add_rtn_u32 random", out, cyc);
    run<1>("ds_add_u32 random (no rtn)", out, cyc);
    run<11>("ds_max_u32 random (no rtn)", out, cyc);
    run<2>("ds_read_b32 random", out, cyc);
    run<3>("ds_read_b64 random", out, cyc);
    run<8>("ds_read_u8 random", out, cyc);
    run<4>("ds_write_b32 random", out, cyc);
    run<5>("ds_write_b64 random", out, cyc);
    run<9>("ds_write_b8 random", out, cyc);
    run<10>("ds_read_b32 linear", out, cyc);
    return 0;
}
