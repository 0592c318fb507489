// Does a pageable H2D pay per SOURCE BUFFER (the runtime pinning it) or once per process (staging set-up)?  Round 4.
//   hipcc --offload-arch=gfx950 -O3 -pthread tools/microbench/exp_h2d_fresh.hip -o tools/microbench/bin/exp_h2d_fresh
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)1 << 30;
    char* d = nullptr; CK(hipMalloc(&d, N));
    for (int huge = 0; huge < 2; ++huge)
        for (int b = 0; b < 4; ++b) {
            char* p = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (huge) madvise(p, N, MADV_HUGEPAGE);
            memset(p, b + 1, N);
            for (int r = 0; r < 3; ++r) { const double t0 = now(); CK(hipMemcpy(d, p, N, hipMemcpyHostToDevice)); printf("buffer %d (huge pages %d), copy %d: %.1f ms\n", b, huge, r, (now() - t0) * 1e3); }
            munmap(p, N);
        }
    char* h = (char*)malloc(N);
    for (int r = 0; r < 3; ++r) { memset(h, r, N); const double t0 = now(); CK(hipMemcpy(d, h, N, hipMemcpyHostToDevice)); printf("malloc'ed buffer rewritten, copy %d: %.1f ms\n", r, (now() - t0) * 1e3); }
    for (int r = 0; r < 3; ++r) { char* q = (char*)malloc(N); memset(q, r, N); const double t0 = now(); CK(hipMemcpy(d, q, N, hipMemcpyHostToDevice)); printf("new malloc each time, copy %d: %.1f ms (%p)\n", r, (now() - t0) * 1e3, (void*)q); free(q); }
    // D2H into fresh vs reused destination
    for (int r = 0; r < 3; ++r) { char* q = (char*)malloc(N); memset(q, r, N); const double t0 = now(); CK(hipMemcpy(q, d, N, hipMemcpyDeviceToHost)); printf("D2H into a touched new malloc, copy %d: %.1f ms\n", r, (now() - t0) * 1e3); free(q); }
    return 0;
}
