// What does the host side of a host-pointer build cost on this box?  (round 4)
// H2D of a 1 GiB text and D2H of a 4 GiB suffix array through every route the runtime offers: pageable hipMemcpy, pinned,
// hipHostRegister'ed caller memory, and a pinned bounce ring emptied by T host threads into FRESHLY allocated memory (the
// result of make_suffix_array is a new std::vector: its pages are touched for the first time by whoever writes them).
//   hipcc --offload-arch=gfx950 -O3 -pthread tools/microbench/exp_host_xfer.hip -o tools/microbench/bin/exp_host_xfer
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void* fresh(size_t bytes, bool huge)
{
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(1); }
    if (huge) madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

// ring: device -> pinned slots (one stream), T threads empty the slots into dst
static double ring_d2h(const char* d_src, char* dst, size_t bytes, int T, size_t chunk, int slots, char* pinned, hipStream_t st)
{
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    std::vector<hipEvent_t> ev(slots);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<std::atomic<long>> slot_chunk(slots);      // chunk whose data the slot holds (issued), -1 free
    std::vector<std::atomic<int>> slot_free(slots);
    for (int s = 0; s < slots; ++s) { slot_chunk[s] = -1; slot_free[s] = 1; }
    std::atomic<long> next_take{0};
    std::atomic<long> issued{0};
    const double t0 = now();
    std::vector<std::thread> th;
    for (int w = 0; w < T; ++w)
        th.emplace_back([&, w] {
            CK(hipSetDevice(0));
            for (;;) {
                const long c = next_take.fetch_add(1);
                if (c >= (long)nchunks) return;
                const int s = (int)(c % slots);
                while (issued.load(std::memory_order_acquire) <= c) std::this_thread::yield();
                CK(hipEventSynchronize(ev[s]));
                const size_t off = (size_t)c * chunk, len = std::min(chunk, bytes - off);
                memcpy(dst + off, pinned + (size_t)s * chunk, len);
                slot_free[s].store(1, std::memory_order_release);
            }
        });
    for (size_t c = 0; c < nchunks; ++c) {
        const int s = (int)(c % slots);
        while (!slot_free[s].load(std::memory_order_acquire)) std::this_thread::yield();
        slot_free[s].store(0);
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        CK(hipMemcpyAsync(pinned + (size_t)s * chunk, d_src + off, len, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev[s], st));
        issued.store((long)c + 1, std::memory_order_release);
    }
    for (auto& t : th) t.join();
    const double dt = now() - t0;
    for (auto& e : ev) CK(hipEventDestroy(e));
    return dt;
}

// ring: T threads fill pinned slots from pageable src, one stream sends them
static double ring_h2d(char* d_dst, const char* src, size_t bytes, int T, size_t chunk, int slots, char* pinned, hipStream_t st)
{
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    std::vector<hipEvent_t> ev(slots);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<std::atomic<long>> filled(slots);          // chunk index + 1 whose data is in the slot
    std::vector<std::atomic<long>> sent(slots);            // chunk index + 1 whose copy was issued from the slot
    for (int s = 0; s < slots; ++s) { filled[s] = 0; sent[s] = 0; }
    std::atomic<long> next_take{0};
    const double t0 = now();
    std::vector<std::thread> th;
    for (int w = 0; w < T; ++w)
        th.emplace_back([&] {
            CK(hipSetDevice(0));
            for (;;) {
                const long c = next_take.fetch_add(1);
                if (c >= (long)nchunks) return;
                const int s = (int)(c % slots);
                if (c >= slots) {          // the slot's previous copy must have left
                    while (sent[s].load(std::memory_order_acquire) != c - slots + 1) std::this_thread::yield();
                    CK(hipEventSynchronize(ev[s]));
                }
                const size_t off = (size_t)c * chunk, len = std::min(chunk, bytes - off);
                memcpy(pinned + (size_t)s * chunk, src + off, len);
                filled[s].store(c + 1, std::memory_order_release);
            }
        });
    for (size_t c = 0; c < nchunks; ++c) {
        const int s = (int)(c % slots);
        while (filled[s].load(std::memory_order_acquire) != (long)c + 1) std::this_thread::yield();
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        CK(hipMemcpyAsync(d_dst + off, pinned + (size_t)s * chunk, len, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(ev[s], st));
        sent[s].store((long)c + 1, std::memory_order_release);
    }
    CK(hipStreamSynchronize(st));
    for (auto& t : th) t.join();
    const double dt = now() - t0;
    for (auto& e : ev) CK(hipEventDestroy(e));
    return dt;
}

int main(int argc, char** argv)
{
    const size_t NT = argc > 1 ? strtoull(argv[1], nullptr, 0) : ((size_t)1 << 30);      // text bytes
    const size_t NS = 4 * NT;                                                              // suffix-array bytes
    CK(hipSetDevice(0));
    { FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { if (fgets(b, sizeof b, f)) printf("THP enabled: %s", b); fclose(f); } }
    { FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/defrag", "r"); char b[128] = {0}; if (f) { if (fgets(b, sizeof b, f)) printf("THP defrag: %s", b); fclose(f); } }
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    char *d_text = nullptr, *d_sa = nullptr;
    CK(hipMalloc(&d_text, NT)); CK(hipMalloc(&d_sa, NS));
    CK(hipMemset(d_sa, 0x5a, NS));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto gbs = [](size_t b, double s) { return b / s / 1e9; };

    // ---- host-only costs
    { char* p = (char*)fresh(NS, false); double t0 = now(); memset(p, 0, NS); double dt = now() - t0; printf("memset of fresh %zu MiB, 1 thread (what std::vector<int32_t>(n+1) pays): %.1f ms = %.1f GB/s\n", NS >> 20, dt * 1e3, gbs(NS, dt));
      t0 = now(); memset(p, 1, NS); dt = now() - t0; printf("memset again (pages present): %.1f ms = %.1f GB/s\n", dt * 1e3, gbs(NS, dt)); munmap(p, NS); }
    for (int huge = 0; huge < 2; ++huge)
        for (int T : {4, 8, 16, 32, 64}) {
            char* p = (char*)fresh(NS, huge);
            const double t0 = now();
            std::vector<std::thread> th;
            for (int w = 0; w < T; ++w) th.emplace_back([&, w] { const size_t a = NS / T * w, b = w == T - 1 ? NS : NS / T * (w + 1); for (size_t o = a; o < b; o += 4096) p[o] = 1; });
            for (auto& t : th) t.join();
            const double dt = now() - t0;
            printf("first touch of %zu MiB (one byte per 4 KiB), %2d threads, MADV_HUGEPAGE %d: %.1f ms = %.1f GB/s\n", NS >> 20, T, huge, dt * 1e3, gbs(NS, dt));
            munmap(p, NS);
        }

    // ---- H2D of the text
    {
        char* src = (char*)fresh(NT, false); memset(src, 3, NT);
        for (int r = 0; r < 2; ++r) { const double t0 = now(); CK(hipMemcpy(d_text, src, NT, hipMemcpyHostToDevice)); const double dt = now() - t0; printf("H2D %zu MiB pageable hipMemcpy (run %d): %.1f ms = %.1f GB/s\n", NT >> 20, r, dt * 1e3, gbs(NT, dt)); }
        { double t0 = now(); CK(hipHostRegister(src, NT, hipHostRegisterDefault)); double tr = now() - t0; t0 = now(); CK(hipMemcpyAsync(d_text, src, NT, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double tc = now() - t0;
          t0 = now(); CK(hipHostUnregister(src)); double tu = now() - t0; printf("H2D %zu MiB hipHostRegister %.1f ms + copy %.1f ms (%.1f GB/s) + unregister %.1f ms\n", NT >> 20, tr * 1e3, tc * 1e3, gbs(NT, tc), tu * 1e3); }
        char* pin = nullptr; CK(hipHostMalloc((void**)&pin, NT, hipHostMallocDefault)); memset(pin, 1, NT);
        for (int r = 0; r < 2; ++r) { const double t0 = now(); CK(hipMemcpyAsync(d_text, pin, NT, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); const double dt = now() - t0; printf("H2D %zu MiB pinned (run %d): %.1f ms = %.1f GB/s\n", NT >> 20, r, dt * 1e3, gbs(NT, dt)); }
        for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20})
            for (int T : {2, 4, 8}) {
                const int slots = 8;
                const double dt = ring_h2d(d_text, src, NT, T, chunk, slots, pin, st);
                printf("H2D %zu MiB pageable through a pinned ring (%d x %zu MiB slots, %d threads): %.1f ms = %.1f GB/s\n", NT >> 20, slots, chunk >> 20, T, dt * 1e3, gbs(NT, dt));
            }
        CK(hipHostFree(pin)); munmap(src, NT);
    }

    // ---- D2H of the suffix array
    {
        { char* dst = (char*)fresh(NS, false); double t0 = now(); CK(hipMemcpy(dst, d_sa, NS, hipMemcpyDeviceToHost)); double dt = now() - t0; printf("D2H %zu MiB into FRESH pageable memory, hipMemcpy: %.1f ms = %.1f GB/s\n", NS >> 20, dt * 1e3, gbs(NS, dt));
          t0 = now(); CK(hipMemcpy(dst, d_sa, NS, hipMemcpyDeviceToHost)); dt = now() - t0; printf("D2H again (pages present): %.1f ms = %.1f GB/s\n", dt * 1e3, gbs(NS, dt)); munmap(dst, NS); }
        for (int huge = 0; huge < 2; ++huge) {
            char* dst = (char*)fresh(NS, huge); double t0 = now(); CK(hipHostRegister(dst, NS, hipHostRegisterDefault)); double tr = now() - t0; t0 = now(); CK(hipMemcpyAsync(dst, d_sa, NS, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double tc = now() - t0;
            t0 = now(); CK(hipHostUnregister(dst)); double tu = now() - t0; printf("D2H %zu MiB fresh memory (MADV_HUGEPAGE %d): hipHostRegister %.1f ms + copy %.1f ms (%.1f GB/s) + unregister %.1f ms\n", NS >> 20, huge, tr * 1e3, tc * 1e3, gbs(NS, tc), tu * 1e3); munmap(dst, NS); }
        const size_t ring_bytes = (size_t)512 << 20;
        char* pin = nullptr; CK(hipHostMalloc((void**)&pin, NS, hipHostMallocDefault)); memset(pin, 1, NS);
        for (int r = 0; r < 2; ++r) { const double t0 = now(); CK(hipMemcpyAsync(pin, d_sa, NS, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); const double dt = now() - t0; printf("D2H %zu MiB pinned (run %d): %.1f ms = %.1f GB/s\n", NS >> 20, r, dt * 1e3, gbs(NS, dt)); }
        for (int huge = 0; huge < 2; ++huge)
            for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20})
                for (int T : {4, 8, 16, 32}) {
                    const int slots = (int)std::min<size_t>(ring_bytes / chunk, (size_t)(4 * T));
                    char* dst = (char*)fresh(NS, huge);
                    const double dt = ring_d2h(d_sa, dst, NS, T, chunk, slots, pin, st);
                    bool ok = true; for (size_t o = 0; o < NS; o += 999983) ok = ok && dst[o] == 0x5a;
                    printf("D2H %zu MiB into FRESH memory (MADV_HUGEPAGE %d) through a pinned ring (%d x %zu MiB, %d threads): %.1f ms = %.1f GB/s%s\n", NS >> 20, huge, slots, chunk >> 20, T, dt * 1e3, gbs(NS, dt), ok ? "" : "  DATA WRONG");
                    munmap(dst, NS);
                }
        // both directions at once (does a D2H stream slow down under an H2D stream?)
        { char* pin2 = nullptr; CK(hipHostMalloc((void**)&pin2, NT, hipHostMallocDefault)); memset(pin2, 1, NT); hipStream_t st2; CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
          const double t0 = now(); CK(hipMemcpyAsync(pin, d_sa, NS, hipMemcpyDeviceToHost, st)); for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(d_text, pin2, NT, hipMemcpyHostToDevice, st2)); CK(hipStreamSynchronize(st)); const double d1 = now() - t0; CK(hipStreamSynchronize(st2)); const double d2 = now() - t0;
          printf("full duplex: D2H %zu MiB %.1f ms (%.1f GB/s) while H2D 4 x %zu MiB %.1f ms (%.1f GB/s)\n", NS >> 20, d1 * 1e3, gbs(NS, d1), NT >> 20, d2 * 1e3, gbs(4 * NT, d2)); CK(hipHostFree(pin2)); }
        CK(hipHostFree(pin));
    }
    return 0;
}
