// Microbenchmark of the 16-bit histogram kernels (half-range u32 vs packed u16).  Not part of the product.
#include "../../msufsort_amd/csrc/sa_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static u64 sm(u64& s) { u64 z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main(int argc, char** argv)
{
    const u64 n = argc > 1 ? strtoull(argv[1], 0, 0) : (1ull << 30) - 1;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;       // 0 random, 1 all 'A', 2 skewed (75 % one pair), 3 period 3
    std::vector<u8> h(n + 64, 0);
    u64 s = 12345;
    for (u64 i = 0; i < n; i += 8) { u64 r = sm(s); memcpy(&h[i], &r, std::min<u64>(8, n - i)); }
    if (kind == 1) memset(h.data(), 'A', n);
    if (kind == 2) for (u64 i = 0; i < n; ++i) if ((h[i] & 3) != 0) h[i] = 'e';
    if (kind == 3) for (u64 i = 0; i < n; ++i) h[i] = "abc"[i % 3];
    u8* d; CK(hipMalloc(&d, n + 64)); CK(hipMemcpy(d, h.data(), n + 64, hipMemcpyHostToDevice));
    u32 *p1, *h1; CK(hipMalloc(&p1, 256ull * 65536 * 4)); CK(hipMalloc(&h1, 65536 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist16<0>), hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u32 m = (u32)n;
    u32 nc = (u32)std::min<u64>(128, std::max<u64>(1, (m + 65535ull) / 65536));
    u64 cl = ((u64)m + nc - 1) / nc; cl = (cl + 32767) / 32768 * 32768;
    const u32 per = cl >= 131072 ? 2 : 1;
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        k_hist16<0><<<nc * per, 1024, H16_LDS_BYTES>>>(d, m, (u32)(cl / per), nc * per, p1, 0u, (const unsigned short*)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    k_reduce16<<<256, 256>>>(p1, nc * per, h1);
    CK(hipDeviceSynchronize());
    std::vector<u32> a(65536, 0), b(65536);
    for (u64 i = 0; i < n; ++i) a[(h[i] << 8) | h[i + 1]]++;
    CK(hipMemcpy(b.data(), h1, 65536 * 4, hipMemcpyDeviceToHost));
    u64 bad = 0; for (int i = 0; i < 65536; ++i) bad += a[i] != b[i];
    printf("n=%llu kind=%d k_hist16 %.3f ms (%.0f GB/s, %.1f %% of 8 TB/s)  mismatches vs host %llu\n", (unsigned long long)n, kind, best, n / best / 1e6, n / best / 1e6 / 80.0, (unsigned long long)bad);
    return 0;
}
