// Host driver + C-ABI of the MI355X suffix-array / BWT engine (see include/msufsort_hip.h).
// Replaces, behind the reference's own API, msufsort::make_suffix_array (reference
// src/library/msufsort/msufsort.cpp:1730-1767), forward_burrows_wheeler_transform (cpp:1771-1817),
// reverse_burrows_wheeler_transform (cpp:1821-2096) and the demo's LCP (main.cpp:16-159).
// There is NO CPU fallback in this library: without a HIP device every entry point fails.
#include "sa_kernels.hip.h"
#include "bwt_kernels.hip.h"
#include "induce_kernels.hip.h"
#include "../../include/msufsort_hip.h"

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));    \
            return MSUFSORT_HIP_ERR_HIP;                                                       \
        }                                                                                      \
    } while (0)

#define TRY(expr)                         \
    do {                                  \
        int r_ = (expr);                  \
        if (r_ != MSUFSORT_HIP_OK) return r_; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return MSUFSORT_HIP_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        need = (need + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); p = nullptr; return MSUFSORT_HIP_ERR_NOMEM; }
        bytes = need;
        return MSUFSORT_HIP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct HostRing;      // pinned bounce ring for device-to-host copies into pageable memory (host_xfer.inc)
struct Copier;        // thread that sends finished row ranges to the host while the build goes on (host_xfer.inc)
struct HostTrace;     // MSUFSORT_HIP_HOST_TRACE timeline (host_xfer.inc)

}  // namespace

// Diagnostic switches of the environment, read ONCE per entry-point call (Switches::load at the top of a
// build, not in its rounds) - they are test hooks and A/B levers, not part of the C-ABI.
struct Switches {
    bool no_refine = false, safe_rank = false, bucket_fast2 = false, no_fast = false, force_fast = false, no_chains = false;
    bool force_retry = false, no_pack = false, no_fuse = false, ind_classic = false, no_ring = false, mid_single = false, no_pcw = false;
    int text_rounds = 0;         // MSUFSORT_HIP_TEXT_ROUNDS (0: unset)
    int digit_bits = 24;         // MSUFSORT_HIP_DIGIT_BITS
    int two_stage = 0; bool two_stage_set = false;      // MSUFSORT_HIP_TWO_STAGE overrides opts->two_stage
    int lcp_cap = 512;           // MSUFSORT_HIP_LCP_CAP
    int ind_spin = 1 << 22;      // MSUFSORT_HIP_IND_SPIN: bound of the induction's look-back spins
    int ind_grid = 0;            // MSUFSORT_HIP_IND_GRID: workgroups per induction level launch (0: what the chip holds at once; tests: a small grid,
                                 // or one larger than the chip holds so that the launch settles on the ticket counter)
    int radix17 = 0;             // MSUFSORT_HIP_RADIX17: -1 never, 0 by size and spread (build_sa), 1 always, after the 16-bit histogram, 2 always, 17-bit
                                 // histogram first (1, 2: test hooks)
    int sync_debug = 0;          // MSUFSORT_HIP_SYNC_DEBUG
    int key1 = 0;                // MSUFSORT_HIP_KEY1: -1 the first gather round always gathers; 0 small alphabets get its key from k_scatter0 when the caller has
                                 // seen few byte values; 1 whenever the alphabet turns out small
    bool host_trace = false;     // MSUFSORT_HIP_HOST_TRACE: timeline of the host-pointer entry points on stderr
    bool no_early_b = false;     // MSUFSORT_HIP_NO_EARLY_B: a host-pointer two-stage build sends rows from its last pass only (A / B)
    bool no_bwt_ride = false;    // MSUFSORT_HIP_NO_BWT_RIDE: the forward transform gathers its bytes after the build (A / B)
    bool no_tiny2 = false;       // MSUFSORT_HIP_NO_TINY2: k_sort_tiny ranks by one key per gather (A / B)
    bool ind_pc_raw = false;     // MSUFSORT_HIP_IND_PC_RAW: the induction's rows carry three plain bytes whatever the alphabet (A / B)
    bool ind_pc_bits5 = false;   // MSUFSORT_HIP_IND_PC_BITS=5: ... 5-bit dense numbers for 17 .. 32 byte values (tests; measured slower)
    int isa_window_kib = 256 << 10;      // MSUFSORT_HIP_ISA_WINDOW_MIB / _KIB (tests): piece of the rank array one pass of its build writes into (0: one pass)
    void load()
    {
        auto on = [](const char* k) { return getenv(k) != nullptr; };
        auto num = [](const char* k, int dflt) { const char* e = getenv(k); return e ? atoi(e) : dflt; };
        no_refine = on("MSUFSORT_HIP_NO_REFINE"); safe_rank = on("MSUFSORT_HIP_SAFE_RANK");
        { const char* e = getenv("MSUFSORT_HIP_BUCKET_SORT"); bucket_fast2 = e && !strcmp(e, "fast2"); }
        no_fast = on("MSUFSORT_HIP_NO_FAST"); force_fast = on("MSUFSORT_HIP_FORCE_FAST"); no_chains = on("MSUFSORT_HIP_NO_CHAINS");
        force_retry = on("MSUFSORT_HIP_FORCE_RETRY"); no_pack = on("MSUFSORT_HIP_NO_PACK"); no_fuse = on("MSUFSORT_HIP_NO_FUSE");
        ind_classic = on("MSUFSORT_HIP_IND_CLASSIC"); no_ring = on("MSUFSORT_HIP_NO_RING"); mid_single = on("MSUFSORT_HIP_MID_SINGLE"); no_pcw = on("MSUFSORT_HIP_NO_PCW");
        text_rounds = std::max(0, num("MSUFSORT_HIP_TEXT_ROUNDS", 0));
        digit_bits = std::min(24, std::max(2, num("MSUFSORT_HIP_DIGIT_BITS", 24)));
        two_stage_set = on("MSUFSORT_HIP_TWO_STAGE"); two_stage = num("MSUFSORT_HIP_TWO_STAGE", 0);
        lcp_cap = std::max(8, num("MSUFSORT_HIP_LCP_CAP", 512));
        ind_spin = std::max(1, num("MSUFSORT_HIP_IND_SPIN", 1 << 22));
        ind_grid = std::max(0, num("MSUFSORT_HIP_IND_GRID", 0));
        radix17 = num("MSUFSORT_HIP_RADIX17", 0);
        sync_debug = num("MSUFSORT_HIP_SYNC_DEBUG", 0);
        key1 = num("MSUFSORT_HIP_KEY1", 0);
        host_trace = on("MSUFSORT_HIP_HOST_TRACE");
        no_early_b = on("MSUFSORT_HIP_NO_EARLY_B");
        no_bwt_ride = on("MSUFSORT_HIP_NO_BWT_RIDE");
        no_tiny2 = on("MSUFSORT_HIP_NO_TINY2");
        ind_pc_raw = on("MSUFSORT_HIP_IND_PC_RAW");
        ind_pc_bits5 = num("MSUFSORT_HIP_IND_PC_BITS", 0) == 5;
        isa_window_kib = std::max(0, on("MSUFSORT_HIP_ISA_WINDOW_KIB") ? num("MSUFSORT_HIP_ISA_WINDOW_KIB", 0) : std::min(1 << 20, num("MSUFSORT_HIP_ISA_WINDOW_MIB", 256)) << 10);
    }
};

// Tied rows of one slice between doubling steps (unordered list of local rows), so that a step costs time in
// proportion to what is still tied, not to the slice.  A list holds at most rows / 16 entries; slices with more tied
// rows are scanned completely (act == nullptr in the kernels) and need the caller's full grp_prev copy instead.
struct ActiveSet {
    DevBuf act[2], prev, cnt;          // cnt: 4 x u64 = {updates, tied rows, next list length, next list overflow}
    u64 count = 0, cap = 0;
    int cur = 0;
    bool valid = false, tried_list = false;
    const void* key_sa = nullptr; u64 key_rows = 0;     // which slice the list describes
    void release() { act[0].release(); act[1].release(); prev.release(); cnt.release(); valid = false; count = 0; cap = 0; }
};

struct msufsort_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;            // second stream of the streaming host path (slices leave while the next shard is sorted)
    hipStream_t gather_stream = nullptr;          // msufsort_hip_forward_bwt_multi: the BWT bytes of a finished slice are gathered (random text reads)
                                                  // while the next shard is sorted (sequential bandwidth) - different bottlenecks, one GPU
    bool attrs_set = false;
    // workspace
    DevBuf rec[3], pool_rec[2], pool_hdr[2];
    DevBuf rec_x[3], pool_x;                      // companions of the records (RecBufs::x): key of the first gather round, small alphabets only
    bool plan_small_alphabet = false;             // what plan_shards saw in the host's copy of the histogram (kept for every shard of the plan)
    bool hint_small_alphabet = false;             // set by the callers that have seen the byte values (tail sample, host histogram): build_sa then
                                                  // lets k_scatter0 produce the companions if the alphabet really has at most 84 codes
    DevBuf lists[2][3], large_round[2], lvl[2], seg0;
    DevBuf alpha, seg0_base, stripe_sums, hist_partial, hist, hist_clip, bstart, child_start, child_cnt, cursor, cursor0, tile_start, trivial, seg_hist;
    DevBuf counters, isa, text_own, sa_own, aux0, aux1, aux2, aux3, doneB, doneC;
    DevBuf h17_partial, h17_fb, h17, child_start17, child_cnt17, cursor17;      // 17-bit radix front end (random-like inputs above the class-C limit)
    u32 h17_q = 1;                                // chunks of k_hist17 per scatter stripe
    std::vector<ActiveSet> active;                // per logical shard (index 0: the per-shard C-ABI pieces)
    // two-stage build (B* sort + induction, induce_host.inc): suffix-type bitmaps, histograms of the B / B* suffixes, sorted
    // B* suffixes, preceding characters of the rows, per-tile counts and the state of the induction passes
    DevBuf ind_sbits, sel_partial, sel_hist, ind_sstar, ind_pc, ind_tiles, ind_state, ind_tables;
    const u8* sel_bits = nullptr;                 // != nullptr: build_sa sorts only the positions whose bit is set
    u32* sel_pc = nullptr;                        // two-stage builds: side array next to the sorted B* suffixes - the characters in front of every suffix
                                                  // whose final row was written by a sort that gathered its key (GatherSpec::pc_out), PC_UNKNOWN elsewhere
    DevBuf ind_spc;
    bool sel_pc_used = false;                     // the last build_sa call wrote to its slice of sel_pc (it then set the whole slice to PC_UNKNOWN first)
    u32* h_ind = nullptr;                         // pinned staging for the induction tables
    u32 ind_resident = 0;                         // workgroups of k_ind_fused the device holds at once (occupancy x CUs), asked once
    DevBuf sub_partial, sub_hist, sub_saved;      // deeper histogram of ONE two-byte key (shard boundaries inside heavy keys)
    int64_t sub_key = -1;                         // which key sub_partial describes (-1: none); valid for the current text only
    // ... and the ones computed before it for the same text (round 6): the plan of a DNA's shards computes the deeper histogram of every
    // straddled key, and every shard build then asked for its boundary keys again - 32 passes of 8.3 ms over the 8 GiB of BASELINE config 5
    struct SubParked { int64_t key = -1; DevBuf partial, hist; };
    std::vector<SubParked> sub_parked;
    void sub_invalidate() { sub_key = -1; for (auto& e : sub_parked) e.key = -1; }
    DevBuf grp_full, grp_prev, upd, upd_cnt;      // single-process sharded builds: tie-group heads (all rows), their copy at the
                                                  // start of a doubling step, rank updates of one row window
    // Histogram computed sharded (multi-GPU jobs, SURVEY 8(e) "Partitioning"; msufsort_hip_hist_part_dev / _hist_plan_dev /
    // _hist_install_dev): this rank counted the stripes [s0, s1) only; the all-reduced totals, the plan made from them and the
    // first-byte sums of ALL stripes over one shard's key range wait here for that shard's build (one use).
    struct SharedHist {
        int stage = 0;               // 1: part counted, 2: planned, 3: stripe sums installed
        const u8* text = nullptr; u64 n = 0, z = 0;
        u32 s0 = 0, s1 = 0;          // my stripes
        u32 per = 1;                 // histogram chunks per stripe in MY partials (finer than the replicated histogram's: the part still fills the chip)
        int n_shards = 0, shard = -1;
        bool small_alphabet = false;
        std::vector<u64> cuts, rows, rank0;
        void reset() { stage = 0; text = nullptr; shard = -1; }
    } xh;
    DevBuf xh_hist, xh_sums;         // 65,536 x u64 totals; [stripes][256] x u32
    // the plan of the last per-shard call (msufsort_hip_make_sa_shard*_dev): with opts->reuse_plan the next shard of the SAME text takes
    // histogram and cuts from it (the histogram's buffers stay as that call left them: only shard calls run in between)
    struct PlanCache {
        bool valid = false; bool wide = false;
        const u8* text = nullptr; u64 n = 0, z = 0; int n_shards = 0;
        bool small_alphabet = false;
        std::vector<u64> cuts, rows, rank0;
        void reset() { valid = false; text = nullptr; }
    } plan_cache;
    bool ext_stripe_sums = false;    // the next run_scan takes xh_sums instead of reducing hist_partial
    u32* h_counters = nullptr;   // pinned
    u64* h_hist = nullptr;       // pinned, 65536 (the 16-bit histogram on the host: shard planning)
    unsigned long long* h_upd = nullptr;   // pinned, 2
    u32 nchunks = 1, chunk_len = 32768;      // text striping of the last k_hist16 (scatter stripes)
    u32 hist_per = 1;                        // histogram chunks per stripe
    u32 list_cap[3] = {0, 0, 0};
    u32 large_cap = 0;
    u64 cap_m = 0;               // records capacity
    u64 cap_for_m = 0;           // largest m the workspace was sized for
    msufsort_hip_timings tm{};
    hipEvent_t ev[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Switches sw;                 // environment switches, reloaded by every entry point
    // rows that are final before the build ends leave for the caller's array at once (msufsort_hip_make_sa_i32_ctx sets the sink;
    // the two-stage build feeds it bucket by bucket during its last pass): sink_rows = rows 1 .. sink_rows handed over so far
    Copier* sink = nullptr;
    int32_t* sink_host = nullptr;
    u64 sink_rows = 0;
    const HostTrace* trace = nullptr;      // the entry point's timeline, for marks from inside the build
    // Forward transform riding on a two-stage build (msufsort_hip_forward_bwt_dev sets d_out): the byte in front of every row is known
    // when the row is (pc[]), so every bucket region's bytes are written on gather_stream as soon as the region is final - beside the
    // remaining induction levels instead of after them - and a host-pointer call (sink) lets them leave at once.
    struct BwtRide {
        u8* d_out = nullptr;          // n bytes on the device, final positions (row r -> r - (r > row of suffix 0))
        int first = -1;               // T[0]: the bucket that holds the row of suffix 0 (-1: ask the device)
        bool done = false;            // every byte is in d_out, the row of suffix 0 in bwt_sent
        Copier* sink = nullptr; u8* host_out = nullptr; u64 sent = 0;      // bytes handed to the copier so far
    } bwt;
    DevBuf bwt_sent;
    HostRing* ring = nullptr;    // created by the first large device-to-host copy of a host-pointer entry point
    std::mutex ring_mu, ring_use;

    template <bool W> int set_mid_attrs()
    {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_mid<CLS_C_THREADS, CLS_C_ITEMS, W>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_mid_lds_bytes<CLS_C_THREADS, CLS_C_ITEMS>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_mid<CLS_B_THREADS, CLS_B_ITEMS, W>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_mid_lds_bytes<CLS_B_THREADS, CLS_B_ITEMS>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_mid<CLS_A_THREADS, CLS_A_ITEMS, W>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_mid_lds_bytes<CLS_A_THREADS, CLS_A_ITEMS>()));
        return MSUFSORT_HIP_OK;
    }

    int set_attrs()
    {
        if (attrs_set) return MSUFSORT_HIP_OK;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist16<0>), hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist16<1>), hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist16<2>), hipFuncAttributeMaxDynamicSharedMemorySize, H16_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan16), hipFuncAttributeMaxDynamicSharedMemorySize, SCAN16_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hist17), hipFuncAttributeMaxDynamicSharedMemorySize, H17_LDS_BYTES));
        TRY(set_mid_attrs<false>());
        TRY(set_mid_attrs<true>());
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_fast2<CLS_C_THREADS, FAST2_C_ITEMS, FAST_BITS_C, FAST2_TL_C>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_fast2_lds_bytes<CLS_C_THREADS, FAST2_C_ITEMS, FAST_BITS_C, FAST2_TL_C>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_fast2<CLS_B_THREADS, CLS_B_ITEMS, FAST_BITS_B, FAST2_TL_B>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_fast2_lds_bytes<CLS_B_THREADS, CLS_B_ITEMS, FAST_BITS_B, FAST2_TL_B>()));
        attrs_set = true;
        return MSUFSORT_HIP_OK;
    }

    // buffers whose size does not depend on the number of suffixes (histogram, offsets, level-0 set-up)
    int ensure_fixed(u32 hist_chunks)
    {
        TRY(seg0.ensure(256 * sizeof(Desc)));
        TRY(child_cnt.ensure(65536 * 4));
        TRY(cursor0.ensure(128 * 256 * 4));
        TRY(seg0_base.ensure(256 * 4));
        TRY(stripe_sums.ensure(128 * 256 * 4));
        TRY(alpha.ensure(256));
        TRY(hist_partial.ensure((size_t)std::max<u32>(hist_chunks, 256u) * 65536 * 4));
        TRY(hist.ensure(65536 * 8));
        TRY(hist_clip.ensure(65536 * 4));
        TRY(bstart.ensure(65537 * 4));
        TRY(counters.ensure(C_NCOUNTERS * 4));
        TRY(child_start.ensure(65536 * 4));
        TRY(cursor.ensure(65536 * 4));
        TRY(tile_start.ensure(257 * 4));
        return MSUFSORT_HIP_OK;
    }

    // buffers that scale with the number of suffixes ONE build sorts (a shard's m, not the input's)
    int ensure_workspace(u64 m)
    {
        TRY(ensure_fixed(256));
        if (cap_for_m >= m) return MSUFSORT_HIP_OK;
        u64 cap = m + m / 4 + (2u << 20);        // + room for the neutral tails of chunked output reservations
        if (cap > 0xfffffff0ull) { set_error("a build of %llu suffixes exceeds the 32-bit record offsets of one shard; use more shards", (unsigned long long)m); return MSUFSORT_HIP_ERR_TOO_LARGE; }
        for (auto& b : rec) TRY(b.ensure(cap * 8));
        for (auto& b : pool_rec) TRY(b.ensure(cap * 8));
        for (auto& b : pool_hdr) TRY(b.ensure(cap * 8));
        list_cap[0] = (u32)(2 * (cap / (TINY_MAX + 1)) + (4u << 20));   // x 2: what the descriptor chunks may give up; + open chunk tails
        list_cap[1] = (u32)(2 * (cap / (CAP_A + 1)) + (4u << 20));
        list_cap[2] = (u32)(2 * (cap / (CAP_B + 1)) + (4u << 20));
        large_cap = (u32)(cap / (CAP_C + 1) + 16);
        for (int s = 0; s < 2; ++s) {
            for (int c = 0; c < 3; ++c) TRY(lists[s][c].ensure((size_t)list_cap[c] * sizeof(Desc)));
            TRY(large_round[s].ensure((size_t)large_cap * sizeof(Desc)));
            TRY(lvl[s].ensure((size_t)large_cap * sizeof(Desc)));
        }
        size_t nchild = std::max<size_t>(65536, (size_t)large_cap * 256);
        TRY(child_start.ensure(nchild * 4));
        TRY(cursor.ensure(nchild * 4));
        TRY(seg_hist.ensure(nchild * 4));
        TRY(tile_start.ensure(((size_t)std::max<u32>(large_cap, 256) + 1) * 4));
        TRY(trivial.ensure((size_t)std::max<u32>(large_cap, 256) * 4));
        TRY(doneB.ensure((size_t)list_cap[1] * 4));
        TRY(doneC.ensure((size_t)list_cap[2] * 4));
        cap_m = cap;
        cap_for_m = m;
        return MSUFSORT_HIP_OK;
    }

    void release_all()
    {
        for (auto& b : rec) b.release();
        for (auto& b : rec_x) b.release();
        pool_x.release();
        for (auto& b : pool_rec) b.release();
        for (auto& b : pool_hdr) b.release();
        for (int s = 0; s < 2; ++s) { for (int c = 0; c < 3; ++c) lists[s][c].release(); large_round[s].release(); lvl[s].release(); }
        alpha.release(); seg0.release(); seg0_base.release(); stripe_sums.release(); hist_partial.release(); hist.release(); hist_clip.release(); bstart.release(); child_start.release(); child_cnt.release();
        cursor.release(); cursor0.release(); tile_start.release(); trivial.release(); seg_hist.release(); counters.release();
        h17_partial.release(); h17_fb.release(); h17.release(); child_start17.release(); child_cnt17.release(); cursor17.release();
        isa.release(); doneB.release(); doneC.release(); text_own.release(); sa_own.release(); aux0.release(); aux1.release(); aux2.release(); aux3.release();
        grp_full.release(); grp_prev.release(); upd.release(); upd_cnt.release(); bwt_sent.release(); xh_hist.release(); xh_sums.release(); xh.reset(); plan_cache.reset();
        sub_partial.release(); sub_hist.release(); sub_saved.release(); sub_key = -1;
        for (auto& e : sub_parked) { e.partial.release(); e.hist.release(); }
        sub_parked.clear();
        ind_sbits.release(); sel_partial.release(); sel_hist.release(); ind_sstar.release(); ind_spc.release();
        ind_pc.release(); ind_tiles.release(); ind_state.release(); ind_tables.release();
        for (auto& a : active) a.release();
        active.clear();
        cap_m = 0; cap_for_m = 0;
    }

    size_t held_bytes() const
    {
        size_t b = 0;
        for (auto& x : rec) b += x.bytes;
        for (auto& x : rec_x) b += x.bytes;
        b += pool_x.bytes;
        for (auto& x : pool_rec) b += x.bytes;
        for (auto& x : pool_hdr) b += x.bytes;
        for (int s = 0; s < 2; ++s) { for (int k = 0; k < 3; ++k) b += lists[s][k].bytes; b += large_round[s].bytes + lvl[s].bytes; }
        for (auto& e : sub_parked) b += e.partial.bytes;
        b += isa.bytes + grp_full.bytes + grp_prev.bytes + upd.bytes + sub_partial.bytes + hist_partial.bytes + seg_hist.bytes + child_start.bytes + cursor.bytes;
        b += doneB.bytes + doneC.bytes + aux0.bytes + aux1.bytes + aux2.bytes + aux3.bytes + sa_own.bytes + text_own.bytes;
        b += ind_sbits.bytes + sel_partial.bytes + ind_sstar.bytes + ind_spc.bytes + ind_pc.bytes + ind_tiles.bytes + h17_partial.bytes;
        for (auto& a : active) b += a.act[0].bytes + a.act[1].bytes + a.prev.bytes;
        return b;
    }

    // frees the sort workspace (records, pools, lists) but keeps the small fixed buffers: single-process builds of very
    // large inputs need the memory for the rank array
    void release_sort_workspace()
    {
        for (auto& b : rec) b.release();
        for (auto& b : pool_rec) b.release();
        for (auto& b : pool_hdr) b.release();
        for (int s = 0; s < 2; ++s) { for (int c = 0; c < 3; ++c) lists[s][c].release(); large_round[s].release(); lvl[s].release(); }
        seg_hist.release(); trivial.release(); doneB.release(); doneC.release();
        cap_m = 0; cap_for_m = 0;
    }

    int read_counters(bool tolerate_overflow = false)
    {
        HIP_TRY(hipMemcpyAsync(h_counters, counters.p, C_NCOUNTERS * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (h_counters[C_ERR] && !tolerate_overflow) { set_error("device capacity check tripped (flags 0x%x)", h_counters[C_ERR]); return MSUFSORT_HIP_ERR_INTERNAL; }
        return MSUFSORT_HIP_OK;
    }
};

namespace {

#include "host_xfer.inc"

__global__ void k_dbg_check_descs(const Desc* list, u32 n, u32 cap, u32 ms, u32* out)
{
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const Desc d = list[i];
        const bool bad = (u64)d.rec_off + d.len > cap || (d.buf & 3u) > 2u || (u64)d.sa_off + d.len > ms;
        if (bad) { const u32 k = atomicAdd(&out[0], 1u); if (k < 6) { out[1 + 5 * k] = i; out[2 + 5 * k] = d.rec_off; out[3 + 5 * k] = d.len; out[4 + 5 * k] = d.sa_off; out[5 + 5 * k] = d.buf; } }
    }
}

__global__ void k_zero_idx(u32* counters, u32 mask)
{
    const u32 t = threadIdx.x;
    if (t < C_NCOUNTERS && ((mask >> t) & 1u)) counters[t] = 0;
}

// counters[dst] = counters[src]
__global__ void k_copy_idx(u32* counters, u32 dst, u32 src) { if (threadIdx.x == 0) counters[dst] = counters[src]; }

__global__ void k_last_nonzero(const u8* __restrict__ text, u64 n, unsigned long long* out)
{
    unsigned long long best = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        if (text[i]) best = i + 1;
    for (int s = 32; s >= 1; s >>= 1) { unsigned long long o = __shfl_xor(best, s, 64); if (o > best) best = o; }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }
inline u32 grid_for(u64 items, u32 per_block = 256, u32 cap = 65536u) { return (u32)std::min<u64>(std::max<u64>((items + per_block - 1) / per_block, 1), cap); }

// number of trailing 0x00 bytes of the device text
int trailing_zeros(msufsort_hip_ctx* c, const u8* d_text, u64 n, u64* z_out,
                   u32* sample_values = nullptr /* distinct byte values among the last 4 KiB and three 1 KiB samples of the body */)
{
    u8 tail[4096], body[3][1024];
    u64 k = std::min<u64>(n, sizeof tail);
    HIP_TRY(hipMemcpyAsync(tail, d_text + (n - k), k, hipMemcpyDeviceToHost, c->stream));
    const bool sample = sample_values && n >= (1u << 20);
    if (sample)
        for (int q = 0; q < 3; ++q) HIP_TRY(hipMemcpyAsync(body[q], d_text + (n / 4) * (q + 1) - 512, 1024, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (sample_values) {
        bool seen[256] = {false};
        u32 d = 0;
        for (u64 i = 0; i < k; ++i) if (!seen[tail[i]]) { seen[tail[i]] = true; ++d; }
        if (sample)
            for (int q = 0; q < 3; ++q)
                for (int i = 0; i < 1024; ++i) if (!seen[body[q][i]]) { seen[body[q][i]] = true; ++d; }
        *sample_values = d;
    }
    u64 z = 0;
    while (z < k && tail[k - 1 - z] == 0) ++z;
    if (z < k || k == n) { *z_out = z; return MSUFSORT_HIP_OK; }
    // the whole tail is zero: scan on the device
    TRY(c->aux0.ensure(8));
    HIP_TRY(hipMemsetAsync(c->aux0.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_last_nonzero, dim3(2048), dim3(256), 0, c->stream, d_text, n, c->aux0.as<unsigned long long>());
    unsigned long long last = 0;
    HIP_TRY(hipMemcpyAsync(&last, c->aux0.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *z_out = n - last;
    return MSUFSORT_HIP_OK;
}

// MSUFSORT_HIP_SYNC_DEBUG=1: synchronise and check after every phase so a faulting kernel is named
#define DBG(label)                                                                                   \
    do {                                                                                             \
        if (c->sw.sync_debug) {                                                                         \
            hipError_t e_ = hipStreamSynchronize(c->stream);                                         \
            if (e_ == hipSuccess) e_ = hipGetLastError();                                            \
            if (e_ != hipSuccess) { set_error("after %s: %s", label, hipGetErrorString(e_)); return MSUFSORT_HIP_ERR_HIP; } \
            if (c->sw.sync_debug > 1) fprintf(stderr, "[dbg] %s ok\n", label);                          \
        }                                                                                            \
    } while (0)

// hist16 + reduce.  Leaves the global 16-bit histogram (u32 narrow / u64 wide) in c->hist.
// scatter stripes: at most 128, each a multiple of 32 KiB; chunks of the 16-bit histogram: a power of two per stripe, so that
// large inputs give every CU a workgroup (one workgroup per chunk, 136 KiB of LDS each) and no chunk exceeds 16 MiB
// (capacity of k_hist16's overflow list)
int plan_stripes(msufsort_hip_ctx* c, u64 m, u32* hchunks_out)
{
    u32 nchunks = (u32)std::min<u64>(128, std::max<u64>(1, (m + 65535) / 65536));
    u64 chunk_len = (m + nchunks - 1) / nchunks;
    chunk_len = (chunk_len + 32767) / 32768 * 32768;
    u32 per = chunk_len >= 131072 ? 2 : 1;
    while (chunk_len / per > (16u << 20)) per *= 2;
    if (chunk_len > 0xffff8000ull) { set_error("input too large for 128 scatter stripes"); return MSUFSORT_HIP_ERR_TOO_LARGE; }
    const u32 hchunks = nchunks * per;
    TRY(c->set_attrs());          // (k_hist16 takes 136 KiB of dynamic LDS; shard planning reaches this before any build)
    TRY(c->ensure_fixed(hchunks));
    c->nchunks = nchunks; c->chunk_len = (u32)chunk_len; c->hist_per = per;
    c->sub_invalidate();
    *hchunks_out = hchunks;
    return MSUFSORT_HIP_OK;
}

template <bool W>
int run_hist(msufsort_hip_ctx* c, const u8* d_text, u64 m)
{
    u32 hchunks = 0;
    c->plan_cache.reset();          // (the histogram buffers a cached shard plan relies on are rewritten here)
    TRY(plan_stripes(c, m, &hchunks));
    const u32 per = c->hist_per;
    const u64 chunk_len = c->chunk_len;
    hipLaunchKernelGGL(k_hist16<0>, dim3(hchunks), dim3(1024), H16_LDS_BYTES, c->stream, d_text, m, (u32)(chunk_len / per), hchunks, c->hist_partial.as<u32>(), 0u, (const unsigned short*)nullptr, 0u);
    hipLaunchKernelGGL(k_reduce16<W>, dim3(256), dim3(256), 0, c->stream, c->hist_partial.as<u32>(), hchunks, c->hist.as<typename Wd<W>::hist_t>());
    HIP_TRY(hipGetLastError());
    return MSUFSORT_HIP_OK;
}

// deeper histogram (next two bytes) of the suffixes that start with the two-byte key `key`; per-chunk partials stay in
// c->sub_partial, the totals (big-endian sub-key order) go to c->sub_hist
template <bool W>
int run_subhist(msufsort_hip_ctx* c, const u8* d_text, u64 m, u32 key)
{
    if (c->sub_key == (int64_t)key) return MSUFSORT_HIP_OK;
    const u32 hchunks = c->nchunks * c->hist_per;
    {   // park the current key's histogram (a handful of them, at most 3 GiB: a DNA has 16 two-byte keys), take this key's if it is parked
        const size_t one = (size_t)hchunks * 65536 * 4;
        if (c->sub_key >= 0 && c->sub_partial.bytes >= one && one <= ((size_t)3 << 30) / 4) {
            const size_t max_parked = std::min<size_t>(16, ((size_t)3 << 30) / one);
            msufsort_hip_ctx::SubParked* slot = nullptr;
            for (auto& e : c->sub_parked) if (e.key < 0) { slot = &e; break; }
            if (!slot && c->sub_parked.size() < max_parked) { c->sub_parked.emplace_back(); slot = &c->sub_parked.back(); }
            if (!slot && !c->sub_parked.empty()) slot = &c->sub_parked[(size_t)c->sub_key % c->sub_parked.size()];      // (all taken: one of them goes)
            if (slot) { std::swap(slot->partial, c->sub_partial); std::swap(slot->hist, c->sub_hist); slot->key = c->sub_key; }
            c->sub_key = -1;
        }
        for (auto& e : c->sub_parked)
            if (e.key == (int64_t)key && e.partial.bytes >= one) {
                std::swap(e.partial, c->sub_partial); std::swap(e.hist, c->sub_hist);
                e.key = -1; c->sub_key = (int64_t)key;
                return MSUFSORT_HIP_OK;
            }
    }
    TRY(c->sub_partial.ensure((size_t)hchunks * 65536 * 4));
    TRY(c->sub_hist.ensure(65536 * 8));
    hipLaunchKernelGGL(k_hist16<1>, dim3(hchunks), dim3(1024), H16_LDS_BYTES, c->stream, d_text, m, c->chunk_len / c->hist_per, hchunks, c->sub_partial.as<u32>(),
                       (key >> 8) | ((key & 255u) << 8), (const unsigned short*)nullptr, 0u);
    hipLaunchKernelGGL(k_reduce16<W>, dim3(256), dim3(256), 0, c->stream, c->sub_partial.as<u32>(), hchunks, c->sub_hist.as<typename Wd<W>::hist_t>());
    HIP_TRY(hipGetLastError());
    c->sub_key = (int64_t)key;
    return MSUFSORT_HIP_OK;
}

// offsets and scatter set-up for the shard that owns the 4-byte prefixes [lo32, hi32)
template <bool W>
int run_scan(msufsort_hip_ctx* c, const u8* d_text, u64 m, u64 lo32, u64 hi32, u64 z, bool from17 = false /* the histograms came from k_hist17 (run_hist17, first) */)
{
    const u32 klo = (u32)(lo32 >> 16), khi = (u32)((hi32 + 0xffffull) >> 16);
    const u32 hchunks = c->nchunks * c->hist_per;
    // (two-stage builds sort a SELECTION of the suffixes: counts and stripe cursors come from the histogram of the selected
    // positions, the alphabet from the histogram of the whole text)
    const typename Wd<W>::hist_t* hist_counts = c->sel_bits ? c->sel_hist.as<typename Wd<W>::hist_t>() : c->hist.as<typename Wd<W>::hist_t>();
    const u32* partial_counts = c->sel_bits ? c->sel_partial.as<u32>() : c->hist_partial.as<u32>();
    hipLaunchKernelGGL(k_hist_clip<W>, dim3(256), dim3(256), 0, c->stream, hist_counts, klo, khi, c->hist_clip.as<u32>(), c->counters.as<u32>());
    const u32* h32 = c->hist_clip.as<u32>();
    // boundary keys owned in part (a shard boundary inside a heavy two-byte key): in-range counts, globally and per chunk
    u32 fix_key[2]; u32 nfix = 0;
    if (khi > klo) {
        const bool flo = (lo32 & 0xffffull) != 0, fhi = (hi32 & 0xffffull) != 0;
        if (flo) fix_key[nfix++] = klo;
        if (fhi && !(flo && khi - 1 == klo)) fix_key[nfix++] = khi - 1;
        if (nfix) TRY(c->sub_saved.ensure((size_t)2 * hchunks * 4));
        for (u32 f = 0; f < nfix; ++f) {
            const u32 bk = fix_key[f];
            const u32 sublo = (bk == klo && flo) ? (u32)(lo32 & 0xffffull) : 0u;
            const u32 subhi = (bk == khi - 1 && fhi) ? (u32)(hi32 & 0xffffull) : 65536u;
            TRY(run_subhist<W>(c, d_text, m, bk));
            HIP_TRY(hipMemsetAsync(c->hist_clip.as<u32>() + bk, 0, 4, c->stream));
            hipLaunchKernelGGL(k_sub_fix, dim3(hchunks), dim3(256), 0, c->stream, c->sub_partial.as<u32>(), sublo, subhi, bk, c->hist_partial.as<u32>(),
                               c->sub_saved.as<u32>() + (size_t)f * hchunks, c->hist_clip.as<u32>() + bk);
            // (a second boundary key reuses the one sub-histogram buffer: stream order keeps this k_sub_fix ahead of its refill)
        }
    }
    hipLaunchKernelGGL(k_scan16, dim3(1), dim3(1024), SCAN16_LDS_BYTES, c->stream, h32, c->bstart.as<u32>(), klo, khi,
                       c->child_start.as<u32>(), c->child_cnt.as<u32>(), c->cursor.as<u32>(), c->seg0_base.as<u32>(),
                       c->seg0.as<Desc>(), c->tile_start.as<u32>(), c->counters.as<u32>(), (u32)z);
    hipLaunchKernelGGL(k_scan16_post, dim3(64), dim3(1024), 0, c->stream, h32, c->bstart.as<u32>(), klo, khi,
                       c->child_start.as<u32>(), c->child_cnt.as<u32>(), c->cursor.as<u32>(), c->counters.as<u32>());
    hipLaunchKernelGGL(k_alphabet<W>, dim3(1), dim3(256), 0, c->stream, c->hist.as<typename Wd<W>::hist_t>(), c->alpha.as<u8>(), c->counters.as<u32>());
    (void)hipMemsetAsync(c->stripe_sums.p, 0, (size_t)c->nchunks * 256 * 4, c->stream);
    if (c->ext_stripe_sums) {      // sharded histogram: the other ranks counted most stripes, the sums came through the all-gather
        c->ext_stripe_sums = false;
        if (from17 || c->sel_bits || nfix) { set_error("stripe sums from a sharded histogram do not fit this build"); return MSUFSORT_HIP_ERR_INTERNAL; }
        HIP_TRY(hipMemcpyAsync(c->stripe_sums.p, c->xh_sums.p, (size_t)c->nchunks * 256 * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    else if (from17) hipLaunchKernelGGL(k_stripe_sums17, dim3(c->nchunks), dim3(256), 0, c->stream, c->h17_fb.as<u32>(), c->h17_q, c->stripe_sums.as<u32>());
    else hipLaunchKernelGGL(k_stripe_sums, dim3(c->nchunks * 4), dim3(256), 0, c->stream, partial_counts, c->hist_per, klo, khi, c->stripe_sums.as<u32>());
    hipLaunchKernelGGL(k_stripes, dim3(256), dim3(128), 0, c->stream, c->stripe_sums.as<u32>(), c->nchunks,
                       c->seg0_base.as<u32>(), c->cursor0.as<u32>());
    for (u32 f = 0; f < nfix; ++f)      // the per-chunk partials serve every shard of this text: put the full counts back
        hipLaunchKernelGGL(k_sub_restore, dim3(cdiv(hchunks, 256)), dim3(256), 0, c->stream, c->hist_partial.as<u32>(), c->sub_saved.as<u32>() + (size_t)f * hchunks, hchunks, fix_key[f]);
    HIP_TRY(hipGetLastError());
    return MSUFSORT_HIP_OK;
}

// 17-bit histogram + offsets of the 131,072 buckets of an UNSHARDED narrow build (k_hist17, k_reduce17, k_scan17): the
// level-1 partition then splits every first-byte segment 512 ways.  Flags (overflow of an 8-bit LDS counter, disagreement
// with the 16-bit histogram) and the largest 17-bit bucket land in counters[C_H17FLAG], counters[C_H17MAX].
// first = true: the 17-bit histogram runs INSTEAD of the 16-bit one (sizes where the 17-bit levels are expected): it also leaves the
// 16-bit histogram (pair sums) in c->hist and the first-byte sums per chunk for the scatter's stripe cursors; the caller has
// planned the stripes (plan_stripes).  first = false: after a 16-bit histogram, cross-checked against it.
int run_hist17(msufsort_hip_ctx* c, const u8* d_text, u64 m, bool first)
{
    // chunks = the stripes of the level-0 scatter cut into q pieces of at most H17_CHUNK bytes (multiples of 16 KiB: a workgroup reads
    // 16 KiB per iteration); q even, so that 128 stripes give a multiple of 256 chunks: whole rounds of one workgroup per CU
    const u32 stripe_len = c->chunk_len;
    c->plan_cache.reset();
    u32 q = std::max<u32>(1, cdiv(stripe_len, H17_CHUNK));
    if (q > 1 && (q & 1u)) ++q;
    const u32 sub_len = (cdiv(stripe_len, q) + 16383u) & ~16383u;
    const u32 nch = c->nchunks * q;
    c->h17_q = q;
    TRY(c->h17_partial.ensure((size_t)nch * 131072));
    TRY(c->h17_fb.ensure((size_t)nch * 256 * 4));
    TRY(c->h17.ensure(131072 * 4));
    TRY(c->child_start17.ensure(131072 * 4));
    TRY(c->child_cnt17.ensure(131072 * 4));
    TRY(c->cursor17.ensure(131072 * 4));
    u32* counters = c->counters.as<u32>();
    HIP_TRY(hipMemsetAsync(c->h17.p, 0, 131072 * 4, c->stream));
    hipLaunchKernelGGL(k_hist17, dim3(nch), dim3(1024), H17_LDS_BYTES, c->stream, d_text, m, stripe_len, q, sub_len, nch, c->h17_partial.as<u32>(), c->h17_fb.as<u32>(),
                       counters + C_H17FLAG);
    const u32 groups = std::max<u32>(1, std::min<u32>(16, nch / 32));
    const u32 per_group = cdiv(nch, groups);
    hipLaunchKernelGGL(k_reduce17, dim3(128, groups), dim3(256), 0, c->stream, c->h17_partial.as<u32>(), nch, per_group, c->h17.as<u32>());
    if (first) hipLaunchKernelGGL(k_pair16, dim3(256), dim3(256), 0, c->stream, c->h17.as<u32>(), c->hist.as<u32>());
    hipLaunchKernelGGL(k_scan17, dim3(128), dim3(1024), 0, c->stream, c->h17.as<u32>(), first ? (const u32*)nullptr : c->hist_clip.as<u32>(), c->child_start17.as<u32>(),
                       c->child_cnt17.as<u32>(), c->cursor17.as<u32>(), counters + C_H17FLAG, counters + C_H17MAX);
    HIP_TRY(hipGetLastError());
    return MSUFSORT_HIP_OK;
}

// Host-only: split the 16-bit key space into n_shards count-balanced contiguous ranges.
// bstart[65537] = exclusive prefix of the 16-bit histogram over the m = n - z radix-sorted suffixes.
void plan_cuts64(const u64* bstart, u64 n, u64 z, int n_shards, u32* cuts, u64* rows)
{
    const u64 m = n - z;
    cuts[0] = 0; rows[0] = 0;
    cuts[n_shards] = 65536; rows[n_shards] = n + 1;
    for (int g = 1; g < n_shards; ++g) {
        const u64 target = (u64)((unsigned __int128)m * (u64)g / (u64)n_shards);
        u32 k = (u32)(std::lower_bound(bstart, bstart + 65536, target) - bstart);
        if (k < cuts[g - 1]) k = cuts[g - 1];
        cuts[g] = k;
        rows[g] = 1 + z + bstart[k];
    }
}

struct ShardCuts {
    std::vector<u64> cuts;       // first 4-byte prefix (big-endian, as a number) of every shard; 2^32 closes the last one
    std::vector<u64> rows;       // first suffix-array row of every shard
    std::vector<u64> rank0;      // global rank (row - 1) of every shard's first radix-sorted suffix
};

// Runs the histogram, brings it to the host and plans the shards.  Leaves the histogram on the device (hist_done).
// A cut that the two-byte keys can only place far behind its balanced target (a key heavier than 1/16 of a shard straddles
// it: DNA has 16 such keys, text has "e ", " t", ...) is refined with the deeper histogram of that key (next two bytes,
// one more pass over the text): the shard boundary then lies inside the key (the reference's analogue is
// largest-partition-first scheduling, cpp:1657-1678).
template <bool W>
int plan_shards(msufsort_hip_ctx* c, const u8* d_text, u64 n, u64 z, int n_shards, ShardCuts& sc)
{
    const u64 m = n - z;
    c->sw.load();
    sc.cuts.assign(n_shards + 1, 0); sc.rows.assign(n_shards + 1, 0); sc.rank0.assign(n_shards + 1, z);
    sc.cuts[n_shards] = 1ull << 32; sc.rows[n_shards] = n + 1;
    if (m == 0) { for (int g = 1; g < n_shards; ++g) { sc.cuts[g] = 1ull << 32; sc.rows[g] = n + 1; } return MSUFSORT_HIP_OK; }
    TRY(run_hist<W>(c, d_text, m));
    HIP_TRY(hipMemcpyAsync(c->h_hist, c->hist.p, 65536 * sizeof(typename Wd<W>::hist_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<u64> bs(65537);
    bs[0] = 0;
    for (u32 k = 0; k < 65536; ++k) bs[k + 1] = bs[k] + (W ? c->h_hist[k] : (u64)reinterpret_cast<const u32*>(c->h_hist)[k]);
    {   // byte values in use (first bytes of the non-empty two-byte keys): the shard builds that follow take it as their hint
        u32 nv = 0;
        for (u32 b = 0; b < 256; ++b) nv += bs[(b + 1) * 256] != bs[b * 256];
        c->plan_small_alphabet = nv <= 83;
    }
    const bool refine = !c->sw.no_refine;
    const u64 tol = std::max<u64>(m / ((u64)n_shards * 16), 1);
    std::vector<u64> sp;             // prefix of the deeper histogram of key `sp_key`
    int64_t sp_key = -1;
    for (int g = 1; g < n_shards; ++g) {
        const u64 target = (u64)((unsigned __int128)m * (u64)g / (u64)n_shards);
        u32 k = (u32)(std::lower_bound(bs.begin(), bs.begin() + 65536, target) - bs.begin());      // first key boundary at or behind the target
        u64 cut = (u64)k << 16, before = bs[k];
        if (refine && k > 0 && bs[k] - target > tol) {
            const u32 kk = k - 1;                                 // the key that straddles the target
            if (sp_key != (int64_t)kk) {
                TRY(run_subhist<W>(c, d_text, m, kk));
                HIP_TRY(hipMemcpyAsync(c->h_hist, c->sub_hist.p, 65536 * sizeof(typename Wd<W>::hist_t), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                sp.assign(65537, 0);
                for (u32 q = 0; q < 65536; ++q) sp[q + 1] = sp[q] + (W ? c->h_hist[q] : (u64)reinterpret_cast<const u32*>(c->h_hist)[q]);
                sp_key = kk;
                if (sp[65536] != bs[kk + 1] - bs[kk]) { set_error("deeper histogram of key %u disagrees with the 16-bit histogram", kk); return MSUFSORT_HIP_ERR_INTERNAL; }
            }
            const u64 want = target - bs[kk];
            const u32 q = (u32)(std::lower_bound(sp.begin(), sp.begin() + 65536, want) - sp.begin());
            cut = ((u64)kk << 16) + q; before = bs[kk] + sp[q];
        }
        if (cut < sc.cuts[g - 1]) { cut = sc.cuts[g - 1]; before = sc.rank0[g - 1] - z; }
        sc.cuts[g] = cut;
        sc.rows[g] = 1 + z + before;
        sc.rank0[g] = z + before;
    }
    sc.rank0[n_shards] = z + m;
    return MSUFSORT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// One round of the engine below the two-byte buckets: partition levels for the large segments, LDS sorts for the
// rest (with the repeat-with-exact-reservations fallback).  Shared by the suffix-array build (text rounds and, for
// narrow single-GPU builds, in-place prefix doubling) and by the stateless doubling step of sharded / wide builds.
// ------------------------------------------------------------------------------------------------
#define MSUFSORT_HIP_UNRESOLVED 1      // (internal status: the build stopped with unresolved tie groups / declined the input)
template <bool W>
struct Rounds {
    typedef typename Wd<W>::sa_t sa_t;
    static constexpr u32 KL = klow<W>();
    msufsort_hip_ctx* c = nullptr;
    hipStream_t st = nullptr;
    u32* counters = nullptr;
    RecBufs bufs{};
    sa_t* sa_local = nullptr;        // row of the shard's first radix-sorted suffix
    u32* isa32 = nullptr;            // MODE_ISA (narrow, in place)
    u32* grp_out = nullptr;          // MODE_DEFER
    int cur = 0;                     // slot of the current round's lists / pool
    u32 sb = 1, nb = 2;              // record buffer holding the current segments / receiving next round's
    u32 mode = MODE_TEXT;
    int round = 0;
    u32 discard = 0;
    u32 klo = 0, khi = 65536;        // two-byte keys the shard touches (skew forecast only)
    u32 cpk = W ? 3u : 4u;
    int verbose = 0;
    bool exact_sticky = false, fast_gave_up = false, force_retry = false, no_pack = false;
    bool safe_rank = false, bucket_sort_bits = true;      // (set from c->sw by init())
    // key of the first gather round from the sequential pass: aux_cand = companion buffers are in place and
    // k_scatter0 / the level-1 partition were given them; aux_on = the alphabet turned out small (<= 84 codes), the companions
    // exist and round 0's kernels carry them; the records round 0 emits then arrive in round 1 WITH their keys
    bool aux_cand = false, aux_on = false;
    void init(msufsort_hip_ctx* ctx, hipStream_t stream, u32* cnt)
    {
        c = ctx; st = stream; counters = cnt;
        safe_rank = c->sw.safe_rank; bucket_sort_bits = !c->sw.bucket_fast2;
        force_retry = c->sw.force_retry; no_pack = c->sw.no_pack;
    }
    u32 nA = 0, nB = 0, nC = 0, nP = 0;
    u32 deep_cap = 0;                    // != 0: k_sort_tiny finishes its runs by comparing the suffixes themselves (two-stage builds)
    GatherSpec gather{nullptr, 0, {}, nullptr};   // text rounds: the sorts (and the first partition level) gather the keys themselves
    const u8* code = nullptr;            // dense alphabet code (device)

    // k_sort_fast needs spread-out keys.  The 16-bit histogram already tells: if its largest bucket is far
    // above the mean the input is text-like and every attempt (reading the records, a serialised LDS-atomic
    // phase, rejection) would be wasted work - go straight to the LSD sort.
    bool keys_spread() const
    {
        const u64 ms_shard = c->h_counters[C_MS];
        return (u64)c->h_counters[C_HMAX] * (u64)(khi - klo) <= 8ull * std::max<u64>(ms_shard, 1);   // max <= 8 x mean
    }
    // Later rounds of a small-alphabet input sort dense base-sigma keys: if the symbols are about evenly used
    // (largest two-byte bucket <= 2 x the mean of the non-empty ones: random DNA, base64, hex dumps) the children of
    // the partition levels are as spread out as random bytes.  The attempt is dropped for the rest of the build as
    // soon as a round hands more than a quarter of its segments back (tandem repeats, text).
    // (round 0's dense digits give class-B children of a few hundred records: the 2^13-entry table of k_sort_fast2 costs
    // more than it saves there - measured on 16..80-symbol random texts - so the dense keys only count from round 1 on)
    // ... and only for alphabets of up to 16 codes (>= 8 symbols per key): measured, a 17-code hex text loses 11 ms to it
    bool wants_fast() const
    {
        const u64 ms_shard = c->h_counters[C_MS];
        const bool dense_uniform = round >= 1 && cpk >= 8u &&
                                   (u64)c->h_counters[C_HMAX] * (u64)std::max<u32>(c->h_counters[C_HNZ], 1u) <= 2ull * std::max<u64>(ms_shard, 1);
        // (k_sort_fast2 exists for narrow records and text keys only)
        return !W && mode == MODE_TEXT && !c->sw.no_fast &&
               (keys_spread() || (dense_uniform && !fast_gave_up) || c->sw.force_fast);
    }

    // tandem repeats (k_chain_resolve): the class lists of `slot` in a doubling round at offset h
    void chain_resolve(int slot, const u32 (&cnt)[3], sa_t* rows, sa_t* isa_rw, const sa_t* isa_ro, u64 n, u64 h)
    {
        if (c->sw.no_chains) return;
        if (cnt[0]) k_chain_resolve<CLS_A_THREADS, CLS_A_ITEMS, W><<<dim3(std::min<u32>(cnt[0], 8192u)), dim3(CLS_A_THREADS), 0, st>>>(
                        bufs, c->lists[slot][0].template as<Desc>(), cnt[0], rows, isa_rw, isa_ro, grp_out, mode, n, h, counters);
        if (cnt[1]) k_chain_resolve<CLS_B_THREADS, CLS_B_ITEMS, W><<<dim3(std::min<u32>(cnt[1], 2048u)), dim3(CLS_B_THREADS), 0, st>>>(
                        bufs, c->lists[slot][1].template as<Desc>(), cnt[1], rows, isa_rw, isa_ro, grp_out, mode, n, h, counters);
        if (cnt[2]) k_chain_resolve<CLS_C_THREADS, CLS_C_ITEMS, W><<<dim3(std::min<u32>(cnt[2], 512u)), dim3(CLS_C_THREADS), 0, st>>>(
                        bufs, c->lists[slot][2].template as<Desc>(), cnt[2], rows, isa_rw, isa_ro, grp_out, mode, n, h, counters);
    }

    Lists make_lists(int slot) const
    {
        Lists L;
        for (int k = 0; k < 3; ++k) { L.cls[k] = c->lists[slot][k].template as<Desc>(); L.cap[k] = c->list_cap[k]; }
        L.cnt_idx = slot ? C_LIST1 : C_LIST0;
        return L;
    }
    u32 cap32() const { return (u32)std::min<u64>(c->cap_m, 0xffffffffu); }

    // children of the 65,536 two-byte buckets (after the level-1 partition of round 0)
    int children_level1(bool radix17 = false)
    {
        hipLaunchKernelGGL(k_children<W>, dim3(radix17 ? 512 : 256), dim3(256), 0, st, bufs, c->seg0.template as<Desc>(), 256u,
                           radix17 ? c->child_start17.template as<u32>() : c->child_start.template as<u32>(), radix17 ? c->child_cnt17.template as<u32>() : c->child_cnt.template as<u32>(),
                           (const u32*)nullptr, 1u, 0u, 2u, radix17 ? 23u : 24u, sa_local, (u32*)nullptr, (u32*)nullptr, (u32)MODE_TEXT,
                           c->pool_rec[cur].template as<u64>(), c->pool_hdr[cur].template as<u64>(), (u32)(cur ? C_POOL1 : C_POOL0), cap32(),
                           make_lists(cur), c->lvl[0].template as<Desc>(), c->large_cap, (u32)C_LVL0, (u32)C_LVLT0, counters, radix17 ? 9u : 8u,
                           aux_cand ? c->pool_x.template as<u32>() : (u32*)nullptr);
        DBG("k_children L1");
        return c->read_counters();
    }

    int levels_and_sorts()
    {
        const int nxt = cur ^ 1;
        u32 a[3];
        { const u32 third = 3 - sb - nb; a[sb] = third; a[third] = sb; a[nb] = nb; }
        if (round == 0) { a[0] = 1; a[1] = 0; a[2] = 2; }   // round 0 ping-pongs between buffers 0 and 1
        // ---- partition levels for large segments ----
        {
            Desc* src_list; u32 nl, ntiles; int lp;       // lp = index of the lvl list used as destination
            u32 shift;
            if (round == 0) {
                src_list = c->lvl[0].template as<Desc>(); nl = c->h_counters[C_LVL0]; ntiles = c->h_counters[C_LVLT0]; lp = 1;
                // Two-byte buckets just above the class-C limit (1.2 .. 2 GiB of random bytes) should not be cut into
                // 256 crumbs of ~100 records each (17 M class-A segments through the slow LSD sort): the level splits
                // on an 8-bit window that starts only `b` bits below the consumed key byte - the rest of the window
                // is constant inside a segment - so children come out around 3/4 of the class-C capacity and go
                // through k_sort_fast2.  Skewed inputs (largest bucket >> class C) keep the full byte.
                u32 b = 1;
                while (b < 8 && ((u64)c->h_counters[C_HMAX] >> b) > (u64)CAP_C * 3 / 4) ++b;
                shift = 24 - b;
            }
            else { src_list = c->large_round[cur].template as<Desc>(); nl = c->h_counters[(cur ? C_LIST1 : C_LIST0) + 3]; ntiles = c->h_counters[cur ? C_LTILES1 : C_LTILES0]; lp = 0; shift = 24; }
            bool first_level = true;
            while (nl > 0) {
                // (small alphabets: k_scatter0 left only the top bits of the key below the bucket byte non-zero; a level that
                // reaches below them ends round 0's splitting)
                const u32 sg0 = c->h_counters[C_ASIGMA];
                const bool packed0 = round == 0 && !no_pack && sg0 >= 2 && sg0 <= 84;
                const u32 bl0 = packed0 ? s0_digit_bits<W>(sg0) : 0u;
                const bool last = (shift <= KL) || (packed0 && shift <= 24 - bl0);
                const u32 cnt_idx = lp ? C_LVL1 : C_LVL0, til_idx = lp ? C_LVLT1 : C_LVLT0;
                hipLaunchKernelGGL(k_zero_idx, dim3(1), dim3(64), 0, st, counters, (1u << cnt_idx) | (1u << til_idx));
                hipLaunchKernelGGL(k_tiles, dim3(1), dim3(1024), 0, st, src_list, nl, c->tile_start.template as<u32>());
                HIP_TRY(hipMemsetAsync(c->seg_hist.p, 0, (size_t)nl * 256 * 4, st));
                HIP_TRY(hipMemsetAsync(c->trivial.p, 0, (size_t)nl * 4, st));
                GatherSpec g0 = gather;
                if (!first_level) g0.text = nullptr;      // (the first level has stored the keys it gathered)
                first_level = false;
                hipLaunchKernelGGL(k_count<W>, dim3(cdiv(ntiles, 4096) * 4096), dim3(P1_THREADS), 0, st, bufs, src_list, nl, c->tile_start.template as<u32>(), shift, c->seg_hist.template as<u32>(), g0, code);
                DBG("k_count");
                hipLaunchKernelGGL(k_segscan, dim3(nl), dim3(256), 0, st, src_list, nl, c->seg_hist.template as<u32>(), c->child_start.template as<u32>(), c->cursor.template as<u32>(), c->trivial.template as<u32>());
                DBG("k_segscan");
                const bool ax = round == 0 && aux_on;          // round 0 of a small alphabet: the records' companions move with them
                if (ax) hipLaunchKernelGGL((k_partition<256, true>), dim3(cdiv(ntiles, 4096) * 4096), dim3(P1_THREADS), 0, st, bufs, src_list, nl, c->tile_start.template as<u32>(), shift,
                                           c->cursor.template as<u32>(), c->trivial.template as<u32>(), a[0], a[1], a[2], counters + C_ASIGMA);
                else hipLaunchKernelGGL(k_partition<256>, dim3(cdiv(ntiles, 4096) * 4096), dim3(P1_THREADS), 0, st, bufs, src_list, nl, c->tile_start.template as<u32>(), shift,
                                   c->cursor.template as<u32>(), c->trivial.template as<u32>(), a[0], a[1], a[2]);
                DBG("k_partition level");
                hipLaunchKernelGGL(k_children<W>, dim3(nl), dim3(256), 0, st, bufs, src_list, nl, c->child_start.template as<u32>(), c->seg_hist.template as<u32>(),
                                   c->trivial.template as<u32>(), a[0], a[1], a[2], shift, sa_local, isa32, grp_out, mode,
                                   c->pool_rec[cur].template as<u64>(), c->pool_hdr[cur].template as<u64>(), (u32)(cur ? C_POOL1 : C_POOL0), cap32(),
                                   make_lists(cur), c->lvl[lp].template as<Desc>(), c->large_cap, cnt_idx, til_idx, counters, 8u,
                                   ax ? c->pool_x.template as<u32>() : (u32*)nullptr);
                DBG("k_children level");
                TRY(c->read_counters());
                src_list = c->lvl[lp].template as<Desc>();
                nl = c->h_counters[cnt_idx];
                ntiles = c->h_counters[til_idx];
                lp ^= 1;
                if (last) {
                    if (nl > 0) {
                        hipLaunchKernelGGL(k_tiles, dim3(1), dim3(1024), 0, st, src_list, nl, c->tile_start.template as<u32>());
                        hipLaunchKernelGGL(k_carry_alloc, dim3(cdiv(nl, 256)), dim3(256), 0, st, src_list, nl, c->trivial.template as<u32>(),
                                           DESC_BUF(32, nb), (u32)(nxt ? C_SEG1 : C_SEG0), cap32(),
                                           c->large_round[nxt].template as<Desc>(), c->large_cap, (u32)((nxt ? C_LIST1 : C_LIST0) + 3), (u32)(nxt ? C_LTILES1 : C_LTILES0), counters);
                        hipLaunchKernelGGL(k_carry_copy<W>, dim3(ntiles), dim3(P1_THREADS), 0, st, bufs, src_list, nl, c->tile_start.template as<u32>(), c->trivial.template as<u32>(),
                                           sa_local, isa32, grp_out, mode, bufs.p[nb], counters, ax ? 1u : 0u);
                        // what the carry reserved must survive a repeated sort attempt (see the retry below)
                        hipLaunchKernelGGL(k_copy_idx, dim3(1), dim3(64), 0, st, counters, (u32)C_CARRY, (u32)(nxt ? C_SEG1 : C_SEG0));
                        DBG("k_carry");
                    }
                    break;
                }
                shift = shift >= 8 + KL ? shift - 8 : KL;
            }
        }
        // ---- LDS sorts of everything that fits ----
        // Persistent workgroups reserve output room in chunks and give up what is left of a chunk that the next
        // request does not fit into; on inputs where every request is about as large as a chunk (e.g. a file followed
        // by a copy of itself: every pool window stays completely tied) that waste can exceed the slack of the
        // buffers.  The sorts only read this round's records, and what they write (rows, ranks, next round's
        // records) is rebuilt identically by a second run, so an overflowing attempt is simply repeated with exact
        // reservations (one global atomic per request, no waste: the live records always fit).
        nA = nB = nC = nP = 0;
        // once a round has overflowed, the following ones start exact while the tied set stays that large
        if (exact_sticky && (u64)c->h_counters[cur ? C_POOL1 : C_POOL0] + c->h_counters[cur ? C_SEG1 : C_SEG0] < c->cap_m / 2) exact_sticky = false;
        for (int attempt = exact_sticky ? 1 : 0; ; ++attempt) {
            Emit em;
            em.pool_rec = c->pool_rec[nxt].template as<u64>(); em.pool_hdr = c->pool_hdr[nxt].template as<u64>();
            em.seg_rec = bufs.p[nb]; em.seg_buf = DESC_BUF(32, nb);
            em.pool_cnt_idx = nxt ? C_POOL1 : C_POOL0; em.seg_cnt_idx = nxt ? C_SEG1 : C_SEG0;
            em.pool_cap = cap32(); em.seg_cap = cap32();
            em.discard = discard; em.grp_out = grp_out;
            em.safe_rank = (attempt > 0 || safe_rank) ? 1u : 0u;
            {   // chunk = what the persistent workgroups reserve per global atomic; slack <= active/16 per kernel
                const u64 act = (u64)c->h_counters[cur ? C_POOL1 : C_POOL0] + c->h_counters[cur ? C_SEG1 : C_SEG0] + (round == 0 ? c->h_counters[C_MS] : 0);
                const u64 ch = std::min<u64>(4096, std::max<u64>(32, act / (16 * 8192)));
                em.pool_chunk = attempt == 0 ? (u32)ch : 0u; em.seg_chunk = em.pool_chunk;
            }
            em.lists = make_lists(nxt);
            // The chunk above is sized for the 8192-workgroup launches (class A, tiny pool).  The class-B / class-C sorts
            // run 1024 / 256 workgroups whose segments emit thousands of records each: with the small chunk nearly every
            // segment goes to the two global counters (one returning atomic per segment, ~90 per us chip-wide), so they
            // take proportionally larger chunks (same total of open chunk tails).
            Emit emB = em, emC = em;
            emB.pool_chunk = emB.seg_chunk = (u32)std::min<u64>(65536, (u64)em.seg_chunk * 8);
            emC.pool_chunk = emC.seg_chunk = (u32)std::min<u64>(262144, (u64)em.seg_chunk * 32);
            const u32 base = cur ? C_LIST1 : C_LIST0;
            nA = c->h_counters[base + 0]; nB = c->h_counters[base + 1]; nC = c->h_counters[base + 2];
            nP = c->h_counters[cur ? C_POOL1 : C_POOL0];
            const bool spread = keys_spread();
            const bool ax0 = round == 0 && aux_on;      // round 0 of a small alphabet: the sorts emit next round's records WITH their keys
            const bool use_fast = wants_fast() && !ax0; // (... which only k_sort_mid / k_sort_tiny know how to do)
            // the companion slot of the LDS sorts has a second use: in the gather rounds of a two-stage build it carries the characters in
            // front of the suffix to the row a record ends up in (GatherSpec::pc_out) - the same instances
            // (class A and the tiny pool only: the larger segments' sorts would pay one more LDS exchange per LSD pass for the few records
            // they finish - measured: +1.1 ms on the 1 GiB text for 6 % of the characters)
            const bool axp = ax0 || (!W && gather.text != nullptr && gather.pc_out != nullptr);
            // k_sort_bits (round 3) where the keys are spread like random bytes; k_sort_fast2 for the dense base-sigma keys of later
            // rounds (random DNA at depth 18: an eighth of the records of a segment tie - more than the dirty list of k_sort_bits
            // holds - and k_sort_fast2 sorts such segments in 2.4 ms per GiB where k_sort_mid takes 3.8) and on request
            // (MSUFSORT_HIP_BUCKET_SORT=fast2)
            const bool use_bits = bucket_sort_bits && spread;
            if (use_fast && (nB || nC)) hipLaunchKernelGGL(k_zero_idx, dim3(1), dim3(64), 0, st, counters, (1u << C_FBB) | (1u << C_FBC));
            if (nC) {
                const u32* ids = nullptr;
                if constexpr (!W) {
                    if (use_fast && use_bits) {
                        k_sort_bits<BITS_C_SHAPE><<<dim3(std::min<u32>(nC, 256u)), dim3(CLS_C_THREADS), 0, st>>>(
                            bufs, c->lists[cur][2].template as<Desc>(), nC, sa_local, em, counters, c->doneC.template as<u32>(), (u32)C_FBC);
                        DBG("k_sort_bits C");
#ifdef BITS_PROF
                        {
                            unsigned long long h[16];
                            hipStreamSynchronize(st);
                            hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bits_prof), sizeof h);
                            fprintf(stderr, "[bits prof] nC=%u cycles/1e6: clear=%.1f A=%.1f scan=%.1f B=%.1f D01=%.1f D2=%.1f D3=%.1f out=%.1f end=%.1f | per segment: dirty %.1f ties %.2f; list overflows %llu, sample len<<32|nl %llx, nl>1700: %llu, nl>2048: %llu\n", nC,
                                    h[0] / 1e6, h[1] / 1e6, h[2] / 1e6, h[3] / 1e6, h[4] / 1e6, h[6] / 1e6, h[7] / 1e6, h[8] / 1e6, h[9] / 1e6,
                                    (double)h[10] / nC, (double)h[11] / nC, h[12], h[13], h[14], h[15]);
                            memset(h, 0, sizeof h);
                            hipMemcpyToSymbol(HIP_SYMBOL(g_bits_prof), h, sizeof h);
                        }
#endif
                        ids = c->doneC.template as<u32>();
                    } else if (use_fast) {
                        k_sort_fast2<CLS_C_THREADS, FAST2_C_ITEMS, FAST_BITS_C, FAST2_TL_C><<<dim3(std::min<u32>(nC, 256u)), dim3(CLS_C_THREADS), sort_fast2_lds_bytes<CLS_C_THREADS, FAST2_C_ITEMS, FAST_BITS_C, FAST2_TL_C>(), st>>>(
                            bufs, c->lists[cur][2].template as<Desc>(), nC, sa_local, isa32, mode, em, counters, c->doneC.template as<u32>(), (u32)C_FBC);
                        DBG("k_sort_fast C");
#ifdef FAST2_PROF
                        {
                            unsigned long long h[16];
                            hipStreamSynchronize(st);
                            hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fast2_prof), sizeof h);
                            fprintf(stderr, "[fast2 prof] nC=%u cycles/1e6:", nC);
                            for (int i = 0; i < 12; ++i) fprintf(stderr, " p%d=%.1f", i, h[i] / 1e6);
                            fprintf(stderr, "\n");
                            memset(h, 0, sizeof h);
                            hipMemcpyToSymbol(HIP_SYMBOL(g_fast2_prof), h, sizeof h);
                        }
#endif
                        ids = c->doneC.template as<u32>();
                    }
                }
                if (ax0) k_sort_mid<CLS_C_THREADS, CLS_C_ITEMS, W, !W><<<dim3(std::min<u32>(nC, 256u)), dim3(CLS_C_THREADS), sort_mid_lds_bytes<CLS_C_THREADS, CLS_C_ITEMS>(), st>>>(
                    bufs, c->lists[cur][2].template as<Desc>(), nC, sa_local, isa32, mode, emC, counters, ids, (u32)C_FBC, gather, code);
                else k_sort_mid<CLS_C_THREADS, CLS_C_ITEMS, W><<<dim3(std::min<u32>(nC, 256u)), dim3(CLS_C_THREADS), sort_mid_lds_bytes<CLS_C_THREADS, CLS_C_ITEMS>(), st>>>(
                    bufs, c->lists[cur][2].template as<Desc>(), nC, sa_local, isa32, mode, emC, counters, ids, (u32)C_FBC, gather, code);
                DBG("k_sort_mid C");
            }
            if (c->sw.sync_debug) {
                for (int k = 0; k < 3; ++k) {
                    const u32 cnt = c->h_counters[base + k];
                    if (!cnt) continue;
                    (void)c->aux0.ensure(256);
                    (void)hipMemsetAsync(c->aux0.p, 0, 256, st);
                    hipLaunchKernelGGL(k_dbg_check_descs, dim3(1024), dim3(256), 0, st, c->lists[cur][k].template as<Desc>(), cnt, cap32(), (u32)c->h_counters[C_MS], c->aux0.template as<u32>());
                    u32 h[40];
                    (void)hipMemcpyAsync(h, c->aux0.p, 160, hipMemcpyDeviceToHost, st);
                    (void)hipStreamSynchronize(st);
                    fprintf(stderr, "[dbg] class %d: %u descriptors, %u bad", k, cnt, h[0]);
                    for (u32 q = 0; q < std::min<u32>(h[0], 6); ++q) fprintf(stderr, " (#%u off %u len %u sa %u buf 0x%x)", h[1 + 5 * q], h[2 + 5 * q], h[3 + 5 * q], h[4 + 5 * q], h[5 + 5 * q]);
                    fprintf(stderr, "\n");
                }
            }
            if (nB) {
                const u32* ids = nullptr;
                if constexpr (!W) {
                    if (use_fast && use_bits) {
                        k_sort_bits<BITS_B_SHAPE><<<dim3(std::min<u32>(nB, 256u * 4u)), dim3(CLS_B_THREADS), 0, st>>>(
                            bufs, c->lists[cur][1].template as<Desc>(), nB, sa_local, em, counters, c->doneB.template as<u32>(), (u32)C_FBB);
                        DBG("k_sort_bits B");
                        ids = c->doneB.template as<u32>();
                    } else if (use_fast) {
                        k_sort_fast2<CLS_B_THREADS, CLS_B_ITEMS, FAST_BITS_B, FAST2_TL_B><<<dim3(std::min<u32>(nB, 256u * 4u)), dim3(CLS_B_THREADS), sort_fast2_lds_bytes<CLS_B_THREADS, CLS_B_ITEMS, FAST_BITS_B, FAST2_TL_B>(), st>>>(
                            bufs, c->lists[cur][1].template as<Desc>(), nB, sa_local, isa32, mode, em, counters, c->doneB.template as<u32>(), (u32)C_FBB);
                        DBG("k_sort_fast B");
                        ids = c->doneB.template as<u32>();
                    }
                }
                if (ax0) k_sort_mid<CLS_B_THREADS, CLS_B_ITEMS, W, !W><<<dim3(std::min<u32>(nB, 1024u)), dim3(CLS_B_THREADS), sort_mid_lds_bytes<CLS_B_THREADS, CLS_B_ITEMS>(), st>>>(
                    bufs, c->lists[cur][1].template as<Desc>(), nB, sa_local, isa32, mode, emB, counters, ids, (u32)C_FBB, gather, code);
                else k_sort_mid<CLS_B_THREADS, CLS_B_ITEMS, W><<<dim3(std::min<u32>(nB, 1024u)), dim3(CLS_B_THREADS), sort_mid_lds_bytes<CLS_B_THREADS, CLS_B_ITEMS>(), st>>>(
                    bufs, c->lists[cur][1].template as<Desc>(), nB, sa_local, isa32, mode, emB, counters, ids, (u32)C_FBB, gather, code);
                DBG("k_sort_mid B");
            }
            // class A (33 .. 512 records): several segments per wave (k_sort_mid_tiles, round 6); MSUFSORT_HIP_MID_SINGLE=1: one segment
            // at a time (the 64-thread instance of k_sort_mid, rounds 2 - 5)
            const u32 gridA = std::min<u32>(cdiv(nA, MIDT_BUN), 8192u);
            if (nA && !c->sw.mid_single && axp) k_sort_mid_tiles<W, !W><<<dim3(gridA), dim3(64), 0, st>>>(
                        bufs, c->lists[cur][0].template as<Desc>(), nA, sa_local, isa32, mode, em, counters, gather, code);
            else if (nA && !c->sw.mid_single) k_sort_mid_tiles<W><<<dim3(gridA), dim3(64), 0, st>>>(
                        bufs, c->lists[cur][0].template as<Desc>(), nA, sa_local, isa32, mode, em, counters, gather, code);
            else if (nA && axp) k_sort_mid<CLS_A_THREADS, CLS_A_ITEMS, W, !W><<<dim3(std::min<u32>(nA, 8192u)), dim3(CLS_A_THREADS), sort_mid_lds_bytes<CLS_A_THREADS, CLS_A_ITEMS>(), st>>>(
                        bufs, c->lists[cur][0].template as<Desc>(), nA, sa_local, isa32, mode, em, counters, (const u32*)nullptr, 0u, gather, code);
            else if (nA) k_sort_mid<CLS_A_THREADS, CLS_A_ITEMS, W><<<dim3(std::min<u32>(nA, 8192u)), dim3(CLS_A_THREADS), sort_mid_lds_bytes<CLS_A_THREADS, CLS_A_ITEMS>(), st>>>(
                        bufs, c->lists[cur][0].template as<Desc>(), nA, sa_local, isa32, mode, em, counters, (const u32*)nullptr, 0u, gather, code);
            DBG("k_sort_mid A");
            if (nP) hipLaunchKernelGGL(k_sort_tiny<W>, dim3(std::min<u32>(cdiv(nP, 256), 8192u)), dim3(256), 0, st, c->pool_rec[cur].template as<u64>(), c->pool_hdr[cur].template as<u64>(),
                                       (u32)(cur ? C_POOL1 : C_POOL0), sa_local, isa32, mode,
                                       em.pool_rec, em.pool_hdr, em.pool_cnt_idx, cap32(), em.pool_chunk, counters, grp_out, discard, gather, code,
                                       (mode == MODE_TEXT && gather.text) ? deep_cap : 0u, ax0 ? c->pool_x.template as<u32>() : (const u32*)nullptr);
            DBG("k_sort_tiny");
            if (round == 0) HIP_TRY(hipEventRecord(c->ev[4], st));
            TRY(c->read_counters(attempt == 0));
#ifdef MID_PROF
            {
                unsigned long long h[3][16];
                hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mid_prof), sizeof h);
                for (int k = 0; k < 3; ++k) {
                    if (!h[k][8]) continue;
                    const double sg = (double)h[k][8];
                    fprintf(stderr, "[mid prof] round %d class %c: %.0f segments seen by thread 0 of each workgroup, %.0f records each; clocks per segment: records %.0f, keys %.0f, sort %.0f, runs+rows %.0f, reserve %.0f, emit %.0f, between %.0f\n",
                            round, "ABC"[k], sg, h[k][9] / sg, h[k][0] / sg, h[k][1] / sg, h[k][2] / sg, h[k][3] / sg, h[k][4] / sg, h[k][5] / sg, h[k][6] / sg);
                }
                memset(h, 0, sizeof h);
                hipMemcpyToSymbol(HIP_SYMBOL(g_mid_prof), h, sizeof h);
            }
#endif
            if (c->h_counters[C_ERR] & 0x400u) return MSUFSORT_HIP_UNRESOLVED;      // a deep comparison gave up: the caller sorts all suffixes
            if (attempt == 0 && force_retry) c->h_counters[C_ERR] |= 0x8000u;      // test hook: MSUFSORT_HIP_FORCE_RETRY=1
            if (c->h_counters[C_ERR] == 0) {
                if (use_fast && !spread && (u64)(c->h_counters[C_FBB] + c->h_counters[C_FBC]) * 4 > (u64)nB + nC) fast_gave_up = true;
                break;
            }
            if (c->h_counters[C_ERR] & ~0x8000u) exact_sticky = true;    // (the test hook repeats EVERY round's first attempt)
            {   // attempt 0 ran out of room: forget what it reserved for the next round and go again
                if (verbose) fprintf(stderr, "[msufsort_hip] round %d: reservation slack exhausted (flags 0x%x), repeating with exact reservations\n", round, c->h_counters[C_ERR]);
                const u32 nP_ = nxt ? C_POOL1 : C_POOL0, nS_ = nxt ? C_SEG1 : C_SEG0, nL_ = nxt ? C_LIST1 : C_LIST0;
                // (carried large segments already sit at the start of next round's segment array, their descriptors in
                // its large list: the segment counter goes back to the end of the carry, not to 0, and the large list stays)
                hipLaunchKernelGGL(k_zero_idx, dim3(1), dim3(64), 0, st, counters, (1u << nP_) | (0x7u << nL_) | (1u << C_ERR));
                hipLaunchKernelGGL(k_copy_idx, dim3(1), dim3(64), 0, st, counters, nS_, (u32)C_CARRY);
            }
        }
        return MSUFSORT_HIP_OK;
    }

    // rotate buffers and slots after a completed round; zero the counters the round after next will fill
    void advance()
    {
        const int nxt = cur ^ 1;
        cur = nxt;
        { const u32 third = 3 - sb - nb; sb = nb; nb = (round == 0) ? 0u : third; if (nb == sb) nb = (sb + 1) % 3; }
        const u32 oP = cur ? C_POOL0 : C_POOL1, oS = cur ? C_SEG0 : C_SEG1, oL = cur ? C_LIST0 : C_LIST1, oT = cur ? C_LTILES0 : C_LTILES1;
        hipLaunchKernelGGL(k_zero_idx, dim3(1), dim3(64), 0, st, counters,
                           (1u << oP) | (1u << oS) | (0xfu << oL) | (1u << oT) | (1u << C_CARRY));
    }
};

// ------------------------------------------------------------------------------------------------
// The suffix-array build for one shard.  d_sa_rows points at row `slice_row_lo` of the suffix array (the shard's first
// row); `with_head` also writes SA[0] and the trailing-zero rows.  rank0 = global rank of the shard's first
// radix-sorted suffix (z + number of suffixes with smaller 16-bit keys).
// d_grp_rows (optional, same indexing as d_sa_rows): receives the LOCAL tie-group head row of every slice row when the
// build stops with unresolved groups (return value MSUFSORT_HIP_UNRESOLVED): sharded builds, and every wide build
// (their prefix doubling is the distributed one further down).
// ------------------------------------------------------------------------------------------------
template <bool W>
int build_sa(msufsort_hip_ctx* c, u8* d_text, u64 n, typename Wd<W>::sa_t* d_sa_rows /* row 0 of this shard's slice */, u64 slice_row_lo,
             u64 z, u64 lo32, u64 hi32, u64 rank0, bool with_head, const msufsort_hip_opts* opts, bool hist_done,
             u32* d_grp_rows = nullptr, u64 slice_rows = 0)
{
    const u32 klo = (u32)(lo32 >> 16), khi = (u32)((hi32 + 0xffffull) >> 16);
    typedef typename Wd<W>::sa_t sa_t;
    const int verbose = opts ? opts->verbose : 0;
    const bool selected = c->sel_bits != nullptr;      // two-stage build: B* suffixes only; deep ties make the caller fall back
    const bool sharded = W || (opts && opts->n_shards > 1) || selected;
    c->sw.load();
    const bool auto_switch = !(opts && opts->text_rounds > 0) && c->sw.text_rounds == 0;
    int text_rounds = (opts && opts->text_rounds > 0) ? opts->text_rounds : 24;
    u64 prev_active = 0;
    if (c->sw.text_rounds > 0) text_rounds = c->sw.text_rounds;
    const u64 m = n - z;
    hipStream_t st = c->stream;
    auto& tm = c->tm;
    memset(&tm, 0, sizeof tm);
    tm.n = (int64_t)n; tm.m = (int64_t)m;
    TRY(c->set_attrs());
    for (auto& a : c->active) { a.valid = false; a.tried_list = false; }      // lists of still-tied rows belong to the previous build's doubling steps
    HIP_TRY(hipEventRecord(c->ev[0], st));
    if (with_head) hipLaunchKernelGGL(k_sa_head<W>, dim3(grid_for(std::max<u64>(z, 1))), dim3(256), 0, st, d_sa_rows, n, z);
    auto finish_groups = [&]() { if (d_grp_rows && slice_rows) hipLaunchKernelGGL(k_grp_iota, dim3(grid_for(slice_rows)), dim3(256), 0, st, d_grp_rows, slice_rows, 0u); };
    const u64 ms = slice_rows ? slice_rows - (with_head ? 1 + z : 0) : m;      // suffixes this build sorts (the shard's, from the planned bounds)
    if (m == 0 || ms == 0) {
        finish_groups();
        HIP_TRY(hipEventRecord(c->ev[5], st)); HIP_TRY(hipStreamSynchronize(st)); return MSUFSORT_HIP_OK;
    }
    TRY(c->ensure_workspace(ms));           // (before anything points into it; growing it reallocates buffers the scan fills)
    u32* counters = c->counters.as<u32>();
    Rounds<W> R;
    R.init(c, st, counters);            // (incl. the test hooks: force_retry throws every round's first sort attempt away)
    R.klo = klo; R.khi = khi; R.verbose = verbose;
    // Key of the first gather round from the sequential pass (narrow records): k_scatter0 holds the text of its
    // tile anyway and writes, next to every record, the key the first gather round would fetch with a random text access per
    // suffix.  Worth it where nearly every suffix is still tied after round 0: small alphabets (text, DNA) - whether this is one
    // the caller has seen (byte values of a tail sample, or the host's histogram); the device decides for good (<= 84 codes).
    const bool hint = c->hint_small_alphabet;
    c->hint_small_alphabet = false;
    c->sel_pc_used = false;
    if constexpr (!W) {
        R.aux_cand = !R.no_pack && !c->sw.force_fast && !c->sw.no_fuse && c->sw.key1 >= 0 && (hint || c->sw.key1 > 0) && ms >= 4096;
        if (R.aux_cand) {
            // (12 bytes per suffix on top of the ~70 of the workspace; if they cannot be had the round simply gathers)
            const size_t xb = (size_t)c->cap_m * 4;
            if (c->rec_x[0].ensure(xb) != MSUFSORT_HIP_OK || c->rec_x[1].ensure(xb) != MSUFSORT_HIP_OK || c->pool_x.ensure(xb) != MSUFSORT_HIP_OK) {
                R.aux_cand = false;
                (void)hipGetLastError();
            }
        }
    }

    // ---- round 0: 16-bit histogram, offsets, two 8-bit scatter levels ----
    // Random-like inputs whose two-byte buckets outgrow the largest LDS sort (uniform bytes from 1.15 GiB): 17 radix bits instead
    // of 16 - a 17-bit histogram of the text (k_hist17) lets the level-1 partition split 512 ways, so that the children
    // fit the bucket sort again instead of passing a third partition level (k_count + k_partition: 24 bytes per suffix more).
    // ... and inputs whose two-byte buckets are just too large for the 4608-record shape (uniform bytes of 290 - 560 MiB): their
    // 17-bit children fill it, where the 18,432-record shape would run a quarter to half empty (3.28 against 3.88 ms at 296 MiB).
    // Unsharded narrow builds only.  At those sizes the 17-bit histogram runs FIRST and yields the 16-bit one as well (its pair
    // sums); whether the keys really are spread out shows afterwards - if not (text, DNA, anything k_hist17 cannot count in its
    // 8-bit LDS counters), the 16-bit histogram runs after all and the build takes the 16-bit levels.
    const u64 mean16 = m >> 16;
    bool radix17 = false, spec17 = false;
    bool cand17 = false;
    if constexpr (!W)
        cand17 = !selected && !(opts && opts->n_shards > 1) && lo32 == 0 && hi32 == (1ull << 32) && c->sw.radix17 >= 0;
    const bool size17 = (double)mean16 + 1.3 * std::sqrt((double)mean16) > (double)CAP_C || (mean16 > (u64)CAP_B && mean16 <= 8900);
    HIP_TRY(hipMemsetAsync(counters, 0, C_NCOUNTERS * 4, st));
    if (cand17 && !hist_done && ((size17 && c->sw.radix17 == 0) || c->sw.radix17 == 2)) {      // (2: test hook - 17 bits first at any size)
        u32 hchunks = 0;
        TRY(plan_stripes(c, m, &hchunks));
        TRY(run_hist17(c, d_text, m, true));
        spec17 = true;
    }
    else if (!hist_done) TRY(run_hist<W>(c, d_text, m));
    HIP_TRY(hipEventRecord(c->ev[1], st));
    TRY(run_scan<W>(c, d_text, m, lo32, hi32, z, spec17));
    R.bufs = RecBufs{{c->rec[0].as<u64>(), c->rec[1].as<u64>(), c->rec[2].as<u64>()},
                     {R.aux_cand ? c->rec_x[0].as<u32>() : nullptr, R.aux_cand ? c->rec_x[1].as<u32>() : nullptr, nullptr}};
    RecBufs& bufs = R.bufs;
    sa_t* sa_local = d_sa_rows + (1 + rank0 - slice_row_lo);
    R.sa_local = sa_local;
    if (spec17 || (cand17 && (size17 || c->sw.radix17 > 0))) {
        TRY(c->read_counters());
        const u64 hmax = c->h_counters[C_HMAX], mean = std::max<u64>(mean16, 1);
        // switch when about a tenth of the two-byte buckets would pass the class-C limit (uniform counts scatter by sqrt(mean): a
        // few oversized buckets are cheaper through one more level of their own than 17 bits for everybody: +20 % at 1120 MiB);
        // buckets just over the 4608-record shape simply take the large one: 17 bits pay once most of them are over
        const bool over_c = hmax > (u64)CAP_C && hmax <= 2ull * CAP_C;
        const bool over_b = mean > (u64)CAP_B && mean <= 8900;
        // (small alphabets: k_scatter0 writes the key symbols as ONE dense base-sigma number below the bucket byte - its top bit is
        // not the 17th bit of the text.  The automatic policy cannot meet them - at most 84^2 of the 65,536 two-byte buckets are
        // in use, so the largest is > 9 x the mean - but the forced one could: round-4 fuzz finding at n = 33)
        const u32 sg17 = c->h_counters[C_ASIGMA];
        const bool packed17 = !R.no_pack && sg17 >= 2 && sg17 <= 84;
        const bool want = !packed17 && (c->sw.radix17 > 0 || ((over_c || over_b) && hmax <= 2 * mean));
        if (spec17) {
            radix17 = want && c->h_counters[C_H17FLAG] == 0;
            if (!radix17) {      // not the input the size promised: the 16-bit histogram after all (k_scatter0's stripes come from its partials)
                if (verbose) fprintf(stderr, "[msufsort_hip] 17-bit histogram first, but the keys are not spread out (flags 0x%x, largest bucket %llu): 16-bit levels\n",
                                     c->h_counters[C_H17FLAG], (unsigned long long)hmax);
                HIP_TRY(hipMemsetAsync(counters, 0, C_NCOUNTERS * 4, st));
                TRY(run_hist<W>(c, d_text, m));
                TRY(run_scan<W>(c, d_text, m, lo32, hi32, z));
                spec17 = false;
            }
        } else if (want) {
            HIP_TRY(hipEventRecord(c->ev[10], st));
            TRY(run_hist17(c, d_text, m, false));
            HIP_TRY(hipEventRecord(c->ev[11], st));
            radix17 = true;
        }
    }
    hipLaunchKernelGGL(k_scatter0<W>, dim3(cdiv(cdiv(m, S0_TILE), 8 * (c->chunk_len / S0_TILE)) * 8 * (c->chunk_len / S0_TILE)), dim3(S0_THREADS), 0, st, d_text, m, lo32, hi32, c->chunk_len, c->cursor0.as<u32>(), bufs.p[0], c->alpha.as<u8>(), counters, R.no_pack ? 0u : 1u, c->sel_bits, bufs.x[0]);
    HIP_TRY(hipEventRecord(c->ev[2], st));
    DBG("k_scatter0");
    if (radix17 && !spec17) {
        TRY(c->read_counters());
        if (c->h_counters[C_H17FLAG]) {
            if (verbose) fprintf(stderr, "[msufsort_hip] 17-bit histogram declined (flags 0x%x): three-level path\n", c->h_counters[C_H17FLAG]);
            radix17 = false;
        }
    }
    if (radix17)
        hipLaunchKernelGGL(k_partition<512>, dim3(cdiv(cdiv(ms, P1_TILE) + 256, 4096) * 4096), dim3(P1_THREADS), 0, st, bufs, c->seg0.as<Desc>(), 256u,
                           c->tile_start.as<u32>(), 23u, c->cursor17.as<u32>(), (const u32*)nullptr, 1u, 0u, 2u);
    else if (R.aux_cand)      // (the kernel looks at the alphabet itself: the host has not seen it yet)
        hipLaunchKernelGGL((k_partition<256, true>), dim3(cdiv(cdiv(ms, P1_TILE) + 256, 4096) * 4096), dim3(P1_THREADS), 0, st, bufs, c->seg0.as<Desc>(), 256u,
                           c->tile_start.as<u32>(), 24u, c->cursor.as<u32>(), (const u32*)nullptr, 1u, 0u, 2u, counters + C_ASIGMA);
    else
        hipLaunchKernelGGL(k_partition<256>, dim3(cdiv(cdiv(ms, P1_TILE) + 256, 4096) * 4096), dim3(P1_THREADS), 0, st, bufs, c->seg0.as<Desc>(), 256u,
                           c->tile_start.as<u32>(), 24u, c->cursor.as<u32>(), (const u32*)nullptr, 1u, 0u, 2u);
    HIP_TRY(hipEventRecord(c->ev[3], st));
    DBG("k_partition L1");
    tm.radix_bits = radix17 ? 17 : 16;

    u64 depth = 5;               // text bytes consumed after round 0 (narrow: bucket bytes 0,1 + key bytes 2,3,4); set below for wide
    KeySpec ks{};                // dense alphabet code of the gather rounds (k_alphabet; known after round 0)
    ks.sigma = 256; ks.cpk = W ? 3u : 4u; ks.zlow = 0; ks.dig_shift = 0; ks.dig_mask = 0xffffffu;

    TRY(R.children_level1(radix17));
    {   // the companions exist iff the alphabet is small (what k_scatter0 and the level-1 partition decided on the device)
        const u32 sg = c->h_counters[C_ASIGMA];
        R.aux_on = R.aux_cand && !radix17 && sg >= 2 && sg <= 84;
        tm.key1_records = R.aux_on ? (int64_t)ms : 0;
        if (verbose && R.aux_cand) fprintf(stderr, "[msufsort_hip] key of the first gather round from the sequential pass: %s (%u codes)\n", R.aux_on ? "yes" : "no", sg);
    }
    if ((u64)c->h_counters[C_MS] != ms && slice_rows) { set_error("slice bounds disagree with the histogram (%u suffixes on the device, %llu planned)", c->h_counters[C_MS], (unsigned long long)ms); return MSUFSORT_HIP_ERR_INTERNAL; }

    if (selected && auto_switch) text_rounds = 64;       // (late rounds of a two-stage build hold a handful of large tie groups: cheap)
    for (;; ++R.round) {
        const int round = R.round;
        // two-stage builds: once the tiny pool is small (or the rounds drag on), its runs are finished by exact suffix
        // comparisons instead of one more key per round - repeated passages agree for thousands of characters
        R.deep_cap = 0;
        if (selected && R.mode == MODE_TEXT && round >= 4) {
            const u64 pool_n = c->h_counters[R.cur ? C_POOL1 : C_POOL0];
            if (pool_n <= ms / 16 || round >= 12) R.deep_cap = 65536u;
        }
        const auto round_t0 = std::chrono::steady_clock::now();
        TRY(R.levels_and_sorts());
        const double round_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - round_t0).count();
        const int nxt = R.cur ^ 1;
        const u32 nb_base = nxt ? C_LIST1 : C_LIST0;
        const u64 actP = c->h_counters[nxt ? C_POOL1 : C_POOL0], actS = c->h_counters[nxt ? C_SEG1 : C_SEG0];
        if (verbose)
            fprintf(stderr, "[msufsort_hip] round %d mode %s depth %llu: sorted A=%u B=%u C=%u tiny=%u -> next tiny=%llu seg=%llu (A=%u B=%u C=%u L=%u), levels + sorts %.2f ms (host clock)\n",
                    round, R.mode == MODE_TEXT ? "text" : "isa", (unsigned long long)depth, R.nA, R.nB, R.nC, R.nP, (unsigned long long)actP, (unsigned long long)actS,
                    c->h_counters[nb_base], c->h_counters[nb_base + 1], c->h_counters[nb_base + 2], c->h_counters[nb_base + 3], round_ms);
        if (verbose && c->h_counters[C_CHAIN]) fprintf(stderr, "[msufsort_hip] round %d: %u suffixes finished as arithmetic progressions (tandem repeats) so far\n", round, c->h_counters[C_CHAIN]);
        if (verbose && R.mode == MODE_TEXT && R.wants_fast() && (c->h_counters[C_FBB] || c->h_counters[C_FBC]))
            fprintf(stderr, "[msufsort_hip] round %d: the bucket sort handed %u class-B and %u class-C segments to k_sort_mid\n", round, c->h_counters[C_FBB], c->h_counters[C_FBC]);
        if (round == 0) tm.unresolved_after_round0 = (int64_t)(actP + actS);
        if (round == 0 && R.wants_fast()) tm.bucket_sort_handed_back = (int64_t)c->h_counters[C_FBB] + (int64_t)c->h_counters[C_FBC];
        if (actP + actS == 0) break;
        if (round == 0) {     // key packing for the gather rounds (k_alphabet ran with the histogram)
            const u32 b = c->h_counters[C_ABITS], sg = c->h_counters[C_ASIGMA];
            const bool packed0 = !R.no_pack && sg >= 2 && sg <= 84;
            if (W) depth = packed0 ? 2 + s0_symbols<W>(sg) : 4;
            {   // (the same function k_scatter0 calls for the companion keys)
                u32 k = 0, zl = 0;
                if (!R.no_pack && key_packing<W>(sg, b, k, zl)) { ks.sigma = sg; ks.cpk = k; ks.zlow = zl; }
            }
            R.cpk = ks.cpk;
            if (verbose) fprintf(stderr, "[msufsort_hip] alphabet: %u codes (%u bits) -> %u symbols per key\n", sg, b, ks.cpk);
        }

        // ---- prepare next round ----
        R.advance();
        const int cur = R.cur;
        const u32 curP = cur ? C_POOL1 : C_POOL0, curS = cur ? C_SEG1 : C_SEG0, curL = cur ? C_LIST1 : C_LIST0;
        // Switch from key gathers to prefix doubling after text_rounds rounds, or (default policy) as soon as the
        // tied set stops shrinking: a round that keeps > 70 % of the previous round's ties means long repeats,
        // where every further 4-byte round is wasted and doubling (log2 LCP rounds) wins despite the ISA build.
        // ... but only once the depth is past what the alphabet needs anyway: ~log_sigma(m) characters separate
        // random suffixes (sigma^2 ~ number of non-empty 16-bit buckets), and a low-entropy alphabet (DNA) keeps
        // everything tied for the first rounds without any repeats being involved.
        const double sigma2 = std::max<double>(4.0, (double)c->h_counters[C_HNZ]);
        const double need = 2.0 * std::log((double)std::max<u64>(m, 2)) / std::log(sigma2);
        // (two-stage builds: a small set that stays tied - a passage with many copies - is cheap to carry through more rounds)
        const bool stalled = auto_switch && (!sharded || selected) && round >= 2 && (double)depth >= std::max(13.0, 1.3 * need) &&
                             (actP + actS) * 10 > prev_active * 7 && !(selected && actP + actS <= ms / 256);
        prev_active = actP + actS;
        if (R.mode == MODE_TEXT && (round + 1 > text_rounds || stalled)) {
            if (sharded) {
                if (selected) { HIP_TRY(hipStreamSynchronize(st)); return MSUFSORT_HIP_UNRESOLVED; }
                if (!d_grp_rows) { set_error("ties deeper than %llu bytes in a sharded build and no group buffer was given", (unsigned long long)depth); return MSUFSORT_HIP_ERR_UNSUPPORTED; }
                // publish the tie groups of my slice (local rows); the caller continues with the distributed doubling
                const u32 row0 = (u32)(1 + rank0 - slice_row_lo);
                u32* grp_local = d_grp_rows + row0;
                hipLaunchKernelGGL(k_grp_iota, dim3(grid_for(std::max<u64>(slice_rows, 1))), dim3(256), 0, st, d_grp_rows, slice_rows, 0u);
                if (actP) hipLaunchKernelGGL(k_grp_pool, dim3(grid_for(actP)), dim3(256), 0, st, c->pool_hdr[cur].as<u64>(), counters, curP, grp_local, row0);
                for (int k = 0; k < 3; ++k) {
                    const u32 cnt = c->h_counters[curL + k];
                    if (cnt) hipLaunchKernelGGL(k_grp_segs, dim3(cnt), dim3(256), 0, st, c->lists[cur][k].as<Desc>(), cnt, grp_local, row0);
                }
                if (u32 cnt = c->h_counters[curL + 3]) hipLaunchKernelGGL(k_grp_segs, dim3(cnt), dim3(256), 0, st, c->large_round[cur].as<Desc>(), cnt, grp_local, row0);
                HIP_TRY(hipEventRecord(c->ev[5], st));
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipGetLastError());
                float ms_ = 0;
                (void)hipEventElapsedTime(&ms_, c->ev[0], c->ev[5]); tm.total_ms = ms_;
                // (the phase clocks of a shard that hands over to the distributed doubling: what build_logical sums into the line of BASELINE config 5)
                (void)hipEventElapsedTime(&ms_, c->ev[0], c->ev[1]); tm.hist16_ms = ms_;
                (void)hipEventElapsedTime(&ms_, c->ev[1], c->ev[2]); tm.scatter0_ms = ms_;
                (void)hipEventElapsedTime(&ms_, c->ev[2], c->ev[3]); tm.scatter1_ms = ms_;
                (void)hipEventElapsedTime(&ms_, c->ev[3], c->ev[4]); tm.bucket_sort_ms = ms_;
                (void)hipEventElapsedTime(&ms_, c->ev[4], c->ev[5]); tm.refine_ms = ms_;
                tm.stop_depth = (int64_t)depth;
                return MSUFSORT_HIP_UNRESOLVED;
            }
            if constexpr (!W) {
                // switch to prefix doubling: build the inverse suffix array
                TRY(c->isa.ensure((size_t)(n + 1) * 4));
                u32* isa = c->isa.as<u32>();
                // ONE random rank write per suffix (round 5; before: every suffix by k_isa_init, then the ~80 % that are still tied AGAIN by
                // k_isa_pool / k_isa_segs - 486 M random 4-byte writes, 15.8 of the 60 ms of a 256 MiB tandem-repeat DNA): the tie-group head
                // of every row is filled in first, with sequential writes (the kernels the sharded exit publishes its groups with), then
                // isa[SA[r]] = 1 + rank0 + head(r) - what the sharded builds do with k_isa_from_slice
                if (c->grp_full.ensure((size_t)std::max<u64>(ms, 1) * 4) == MSUFSORT_HIP_OK) {
                    u32* grp = c->grp_full.as<u32>();
                    hipLaunchKernelGGL(k_grp_iota, dim3(grid_for(ms)), dim3(256), 0, st, grp, ms, 0u);
                    if (actP) hipLaunchKernelGGL(k_grp_pool, dim3(grid_for(actP)), dim3(256), 0, st, c->pool_hdr[cur].as<u64>(), counters, curP, grp, 0u);
                    for (int k = 0; k < 3; ++k) {
                        const u32 cnt = c->h_counters[curL + k];
                        if (cnt) hipLaunchKernelGGL(k_grp_segs, dim3(cnt), dim3(256), 0, st, c->lists[cur][k].as<Desc>(), cnt, grp, 0u);
                    }
                    if (u32 cnt = c->h_counters[curL + 3]) hipLaunchKernelGGL(k_grp_segs, dim3(cnt), dim3(256), 0, st, c->large_round[cur].as<Desc>(), cnt, grp, 0u);
                    // ... a pass per 256 MiB window of the rank array: the random writes of a pass then land in a piece the memory-side
                    // cache holds (70 G writes/s instead of the 27 G/s HBM takes partial lines at); a pass reads the rows again (4 bytes
                    // per suffix: 0.27 ms per GiB).  Measured on tandem-repeat DNA, 2^27 / 2^28 / 2^29 / 2^30 bytes: 29.9 -> 28.2,
                    // 55.4 -> 50.4, 107.1 -> 96.7, 217.6 -> 212.4 ms; 128 and 512 MiB windows lose at every size
                    const u64 wl = (u64)c->sw.isa_window_kib << 8;      // suffixes per window
                    if (wl > 0 && n > wl && n / wl <= 4096) {
                        for (u64 lo = 0; lo < n; lo += wl)
                            hipLaunchKernelGGL(k_isa_from_slice_win, dim3(grid_for(ms)), dim3(256), 0, st, sa_local, grp, ms, (u32)(rank0 + 1), isa, (u32)lo, (u32)std::min<u64>(lo + wl, n));
                    } else
                    hipLaunchKernelGGL(k_isa_from_slice<false>, dim3(grid_for(ms)), dim3(256), 0, st, sa_local, grp, ms, rank0 + 1, isa);
                    if (z) hipLaunchKernelGGL(k_isa_zero_tail, dim3(grid_for(z)), dim3(256), 0, st, isa, (u32)n, (u32)z);
                    DBG("isa from group heads");
                } else {
                (void)hipGetLastError();
                hipLaunchKernelGGL(k_isa_init, dim3(grid_for(m + z)), dim3(256), 0, st, sa_local, counters, isa, (u32)n, (u32)z);
                DBG("k_isa_init");
                if (actP) hipLaunchKernelGGL(k_isa_pool, dim3(grid_for(actP)), dim3(256), 0, st, c->pool_rec[cur].as<u64>(), c->pool_hdr[cur].as<u64>(), counters, curP, isa);
                DBG("k_isa_pool");
                for (int k = 0; k < 3; ++k) {
                    const u32 cnt = c->h_counters[curL + k];
                    if (cnt) hipLaunchKernelGGL(k_isa_segs, dim3(cnt), dim3(256), 0, st, bufs, c->lists[cur][k].as<Desc>(), cnt, counters, isa);
                }
                if (u32 cnt = c->h_counters[curL + 3]) hipLaunchKernelGGL(k_isa_segs, dim3(cnt), dim3(256), 0, st, bufs, c->large_round[cur].as<Desc>(), cnt, counters, isa);
                }
                DBG("isa build");
                R.mode = MODE_ISA;
                R.isa32 = isa;
            }
        }
        // keys of all still-tied suffixes: gathered by the sorts themselves in text rounds (unless k_sort_fast2 will be tried,
        // which wants its records complete), by k_refill otherwise
        ks.depth = depth;
        const sa_t* isa_any = nullptr;
        if constexpr (!W) isa_any = R.isa32;
        R.round = round + 1;              // (wants_fast looks at the round that is about to run)
        // round 0 of a small alphabet emitted its still-tied records WITH the key of this round (k_scatter0 read it off the text
        // it was holding): no gather, neither fused into the sorts nor as a k_refill pass
        const bool keyed = round == 0 && R.aux_on && R.mode == MODE_TEXT;
        const bool fuse = R.mode == MODE_TEXT && !keyed && !R.wants_fast() && !c->sw.no_fuse;
        R.round = round;
        R.code = c->alpha.as<u8>();
        // two-stage builds: the gathering sorts also pick up the characters in front of the suffixes they finish (GatherSpec::pc_out) - in rounds
        // that gather for at least an eighth of the build's suffixes (a DNA's gather rounds hold 1 % of them: not worth a second load per record);
        // the slice is set to PC_UNKNOWN right before the first such round (rows that became final earlier hold nothing yet)
        u32* pc_out = nullptr;
        if (fuse && c->sel_pc && (actP + actS) * 8 >= ms) {
            pc_out = c->sel_pc + (1 + rank0 - slice_row_lo);
            if (!c->sel_pc_used) { HIP_TRY(hipMemsetAsync(pc_out, 0xff, (size_t)ms * 4, st)); c->sel_pc_used = true; }
        }
        // narrow text rounds with keys of up to 8 symbols: k_sort_tiny ranks by this round's key and the next one's (GS_TINY2)
        const u32 gflags = (!W && fuse && ks.cpk <= 8u && !c->sw.no_tiny2) ? GS_TINY2 : 0u;
        R.gather = GatherSpec{fuse ? d_text : nullptr, n, ks, pc_out, gflags};
        // tandem repeats: tie groups that are one arithmetic progression of positions are finished at once (k_chain_resolve)
        if constexpr (!W) {
            if (R.mode == MODE_ISA && depth <= 4096) {
                const u32 cl[3] = {c->h_counters[curL], c->h_counters[curL + 1], c->h_counters[curL + 2]};
                R.chain_resolve(cur, cl, sa_local, R.isa32, (const u32*)nullptr, n, depth);
            }
        }
        if (!fuse && !keyed && actP) hipLaunchKernelGGL(k_refill<W>, dim3(std::min<u32>(cdiv(actP, 1024), 65536u)), dim3(256), 0, st, c->pool_rec[cur].as<u64>(), counters, curP,
                                     d_text, isa_any, n, R.mode, c->alpha.as<u8>(), ks);
        if (!fuse && !keyed && actS) hipLaunchKernelGGL(k_refill<W>, dim3(std::min<u32>(cdiv(actS, 1024), 65536u)), dim3(256), 0, st, bufs.p[R.sb], counters, curS,
                                     d_text, isa_any, n, R.mode, c->alpha.as<u8>(), ks);
        DBG("k_refill");
        if (R.mode == MODE_TEXT) depth += ks.cpk; else { depth *= 2; tm.doubling_rounds++; }
        tm.rounds++;
        if (!keyed) tm.gathered_records += (int64_t)(actP + actS);          // records whose next key was gathered (roofline of the key rounds)
    }
    finish_groups();
    HIP_TRY(hipEventRecord(c->ev[5], st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    tm.progression_suffixes = (int64_t)c->h_counters[C_CHAIN];
    float ms_ = 0;
    (void)hipEventElapsedTime(&ms_, c->ev[0], c->ev[5]); tm.total_ms = ms_;
    (void)hipEventElapsedTime(&ms_, c->ev[0], c->ev[1]); tm.hist16_ms = ms_;
    (void)hipEventElapsedTime(&ms_, c->ev[1], c->ev[2]); tm.scatter0_ms = ms_;
    (void)hipEventElapsedTime(&ms_, c->ev[2], c->ev[3]); tm.scatter1_ms = ms_;
    (void)hipEventElapsedTime(&ms_, c->ev[3], c->ev[4]); tm.bucket_sort_ms = ms_;
    (void)hipEventElapsedTime(&ms_, c->ev[4], c->ev[5]); tm.refine_ms = ms_;
    if (spec17) tm.hist17_ms = tm.hist16_ms;      // the 17-bit histogram ran INSTEAD of the 16-bit one
    else if (tm.radix_bits == 17) {             // ... or between ev[1] and the level-0 scatter: bill it to the histograms
        (void)hipEventElapsedTime(&ms_, c->ev[10], c->ev[11]); tm.hist17_ms = ms_;
        tm.scatter0_ms -= ms_; tm.hist16_ms += ms_;
    }
    return MSUFSORT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Distributed prefix doubling (see the block comment above k_grp_iota in sa_kernels.hip.h).
// ------------------------------------------------------------------------------------------------
struct Digits { u32 npass; u32 shift[2], mask[2]; };

// narrow: the rank is the key.  wide: 24 key bits per pass; ranks up to n need bit_length(n) bits.
template <bool W>
Digits plan_digits(u64 n, int digit_bits)
{
    Digits d{1, {0, 0}, {0xffffffffu, 0}};
    if (!W) return d;
    const u32 kb = (u32)digit_bits;                      // (24; test hook MSUFSORT_HIP_DIGIT_BITS: narrower passes)
    u32 B = 0;
    while (B < 63 && (n >> B) != 0) ++B;
    if (B <= kb) { d.mask[0] = 0xffffffu; return d; }
    const u32 S = std::min<u32>(B - kb, 24u);          // (B <= 48 always holds for 40-bit indices)
    d.npass = 2;
    d.shift[0] = S; d.mask[0] = 0xffffffu;
    d.shift[1] = 0; d.mask[1] = (1u << S) - 1u;
    return d;
}

// One sort pass of a doubling step over the rows of one slice (stateless: everything is rebuilt from rows + grp).
// Returns the number of rows that were still tied BEFORE the pass in *tied_in (0: nothing to do).
template <bool W>
int double_pass(msufsort_hip_ctx* c, u64 n, typename Wd<W>::sa_t* d_sa_slice, u32* d_grp_slice, u64 rows, const u32* act, u32 nact,
                const typename Wd<W>::sa_t* d_isa, u64 h, u32 dig_shift, u32 dig_mask, int verbose, u64* tied_in, bool first_pass)
{
    hipStream_t st = c->stream;
    if (rows > 0xfffffff0ull) { set_error("slice of %llu rows: too large for one shard", (unsigned long long)rows); return MSUFSORT_HIP_ERR_TOO_LARGE; }
    const u32 m = (u32)rows;
    *tied_in = 0;
    if (m < 2) return MSUFSORT_HIP_OK;
    TRY(c->set_attrs());
    TRY(c->ensure_workspace(m));
    u32* counters = c->counters.as<u32>();
    HIP_TRY(hipMemsetAsync(counters, 0, C_NCOUNTERS * 4, st));
    const u32 hv[2] = {m, 0u};
    HIP_TRY(hipMemcpyAsync(counters + C_MS, hv, 8, hipMemcpyHostToDevice, st));      // C_MS, C_RANK0
    Rounds<W> R;
    R.init(c, st, counters);
    R.bufs = RecBufs{{c->rec[0].as<u64>(), c->rec[1].as<u64>(), c->rec[2].as<u64>()}};
    R.sa_local = d_sa_slice; R.grp_out = d_grp_slice;
    R.cur = 0; R.sb = 2; R.nb = 0; R.mode = MODE_DEFER; R.round = 1; R.discard = 1; R.verbose = verbose;
    R.force_retry = false;
    hipLaunchKernelGGL(k_import_groups<W>, dim3(cdiv(act ? std::max<u32>(nact, 1u) : m, 256)), dim3(256), 0, st, d_sa_slice, d_grp_slice, m, act, nact, R.sb,
                       c->pool_rec[0].as<u64>(), c->pool_hdr[0].as<u64>(), (u32)C_POOL0, R.cap32(),
                       R.make_lists(0), c->large_round[0].as<Desc>(), c->large_cap, (u32)(C_LIST0 + 3), (u32)C_LTILES0, counters);
    DBG("k_import_groups");
    TRY(c->read_counters());
    const u64 actP = c->h_counters[C_POOL0];
    const u64 nseg = (u64)c->h_counters[C_LIST0] + c->h_counters[C_LIST0 + 1] + c->h_counters[C_LIST0 + 2] + c->h_counters[C_LIST0 + 3];
    *tied_in = actP + nseg;            // (a count of pool rows + segments: only its being zero matters)
    if (actP + nseg == 0) return MSUFSORT_HIP_OK;
    // tandem repeats: groups that are one arithmetic progression of positions are finished before anything is gathered for
    // them (first digit pass of a step only: the second pass sorts inside the groups the first one left)
    if (first_pass && h <= 4096) {
        const u32 cl[3] = {c->h_counters[C_LIST0], c->h_counters[C_LIST0 + 1], c->h_counters[C_LIST0 + 2]};
        R.chain_resolve(0, cl, d_sa_slice, (typename Wd<W>::sa_t*)nullptr, d_isa, n, h);
        DBG("k_chain_resolve");
    }
    KeySpec ks{};
    ks.depth = h; ks.sigma = 256; ks.cpk = W ? 3u : 4u; ks.zlow = 0; ks.dig_shift = dig_shift; ks.dig_mask = dig_mask;
    if (actP) hipLaunchKernelGGL(k_refill<W>, dim3(std::min<u32>(cdiv(actP, 1024), 65536u)), dim3(256), 0, st, c->pool_rec[0].as<u64>(), counters, (u32)C_POOL0,
                                 (const u8*)nullptr, d_isa, n, (u32)MODE_DEFER, (const u8*)nullptr, ks);
    if (nseg) hipLaunchKernelGGL(k_refill_rows<W>, dim3(std::min<u32>(cdiv(act ? std::max<u32>(nact, 1u) : m, 1024), 65536u)), dim3(256), 0, st, d_sa_slice, d_grp_slice, m, act, nact, R.bufs.p[R.sb], d_isa, n, ks);
    DBG("doubling refill");
    TRY(R.levels_and_sorts());
    if (verbose)
        fprintf(stderr, "[msufsort_hip] doubling pass h=%llu digit>>%u: rows %u, sorted A=%u B=%u C=%u L=%u tiny=%u\n", (unsigned long long)h, dig_shift, m,
                R.nA, R.nB, R.nC, c->h_counters[C_LIST0 + 3], R.nP);
    return MSUFSORT_HIP_OK;
}

// One doubling step for one slice: remember the groups, then one or two sort passes.  *items = work items the emit
// pass must look at afterwards (entries of the active list, or rows).
template <bool W>
int double_sort(msufsort_hip_ctx* c, ActiveSet& as, u64 n, typename Wd<W>::sa_t* d_sa_slice, u32* d_grp_slice, u32* d_grp_prev_slice, u64 rows,
                const typename Wd<W>::sa_t* d_isa, u64 h, int verbose, u64* tied_in, u64* items)
{
    c->sw.load();
    const Digits dg = plan_digits<W>(n, c->sw.digit_bits);
    hipStream_t st = c->stream;
    if (as.key_sa != (const void*)d_sa_slice || as.key_rows != rows) { as.valid = false; as.tried_list = false; as.key_sa = d_sa_slice; as.key_rows = rows; }
    if (!as.cap) {
        as.cap = std::max<u64>(rows / 16, 1024);
        for (auto& b : as.act) TRY(b.ensure(as.cap * 4));
        TRY(as.prev.ensure(as.cap * 4));
        TRY(as.cnt.ensure(32));
    }
    HIP_TRY(hipMemsetAsync(as.cnt.p, 0, 32, st));
    if (!as.valid && !as.tried_list) {
        // first step on this slice: one cheap look at the group heads tells whether the tied rows fit a list
        as.tried_list = true;
        hipLaunchKernelGGL(k_list_tied, dim3(grid_for(rows, 256 * UPD_K, 16384u)), dim3(256), 0, st, d_grp_slice, rows, as.act[as.cur].template as<u32>(), as.cap, as.cnt.template as<unsigned long long>());
        HIP_TRY(hipMemcpyAsync(c->h_upd, as.cnt.p, 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (c->h_upd[3] == 0) { as.valid = true; as.count = c->h_upd[2]; }
        HIP_TRY(hipMemsetAsync(as.cnt.p, 0, 32, st));
    }
    const bool list = as.valid;
    const u32* act = list ? as.act[as.cur].template as<u32>() : nullptr;
    const u32 nact = list ? (u32)as.count : 0u;
    *items = list ? as.count : rows;
    *tied_in = 0;
    if (list && as.count == 0) return MSUFSORT_HIP_OK;
    if (list) hipLaunchKernelGGL(k_gather_prev, dim3(grid_for(as.count)), dim3(256), 0, st, d_grp_slice, act, as.count, as.prev.template as<u32>());
    else HIP_TRY(hipMemcpyAsync(d_grp_prev_slice, d_grp_slice, (size_t)rows * 4, hipMemcpyDeviceToDevice, st));
    u64 t0 = 0;
    for (u32 p = 0; p < dg.npass; ++p) {
        u64 t = 0;
        TRY((double_pass<W>(c, n, d_sa_slice, d_grp_slice, rows, act, nact, d_isa, h, dg.shift[p], dg.mask[p], verbose, &t, p == 0)));
        if (p == 0) t0 = t;
        if (t == 0) break;
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    *tied_in = t0;
    return MSUFSORT_HIP_OK;
}

// rank updates of the work items [i0, i1) of a slice -> d_out (capacity `cap` updates); *count = updates, *tied = rows of
// the window that are still tied.  The window that ends at `items_total` closes the step: the next active list is adopted.
template <bool W>
int emit_updates(msufsort_hip_ctx* c, ActiveSet& as, const typename Wd<W>::sa_t* d_sa_slice, const u32* d_grp_slice, const u32* d_grp_prev_slice, u64 rows, u64 slice_lo,
                 u64 i0, u64 i1, u64 items_total, u64* d_out, u64 cap, u64* count, u64* tied)
{
    hipStream_t st = c->stream;
    if (!as.cap || as.key_sa != (const void*)d_sa_slice) { set_error("emit_updates without a preceding double_sort on this slice"); return MSUFSORT_HIP_ERR_BAD_ARG; }
    const bool list = as.valid;
    unsigned long long* cnt = as.cnt.template as<unsigned long long>();
    HIP_TRY(hipMemcpyAsync(c->h_upd, cnt, 16, hipMemcpyDeviceToHost, st));     // (cnt[0..1] accumulate over the windows of a step)
    HIP_TRY(hipStreamSynchronize(st));
    const u64 c0 = c->h_upd[0], t0 = c->h_upd[1];
    if (i1 > i0) {
        // updates are written from index 0 of d_out in every window: pass the buffer shifted back by what earlier windows counted
        hipLaunchKernelGGL(k_emit_updates<W>, dim3(grid_for(i1 - i0, 256 * UPD_K, 16384u)), dim3(256), 0, st, d_sa_slice, d_grp_slice,
                           list ? as.prev.template as<u32>() : d_grp_prev_slice, list ? as.act[as.cur].template as<u32>() : (const u32*)nullptr,
                           rows, i0, i1, slice_lo, d_out - (W ? 2 : 1) * c0, cap + c0, as.act[as.cur ^ 1].template as<u32>(), as.cap, cnt);
    }
    HIP_TRY(hipMemcpyAsync(c->h_upd, cnt, 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (c->h_upd[0] - c0 > cap) { set_error("rank-update window overflow (%llu > %llu)", c->h_upd[0] - c0, (unsigned long long)cap); return MSUFSORT_HIP_ERR_INTERNAL; }
    *count = c->h_upd[0] - c0; *tied = c->h_upd[1] - t0;
    if (i1 >= items_total) {          // step complete: adopt the list of still-tied rows (if it fitted)
        as.valid = c->h_upd[3] == 0;
        as.count = as.valid ? c->h_upd[2] : 0;
        as.cur ^= 1;
    }
    return MSUFSORT_HIP_OK;
}

template <bool W>
int apply_updates(msufsort_hip_ctx* c, const u64* d_upd, u64 count, typename Wd<W>::sa_t* d_isa)
{
    if (count) hipLaunchKernelGGL(k_apply_updates<W>, dim3(grid_for(count, 256, 65536u)), dim3(256), 0, c->stream, d_upd, count, d_isa);
    HIP_TRY(hipGetLastError());
    return MSUFSORT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Single-process build with G logical shards on one GPU: what the int64 entry points run for inputs beyond 2^31 - 2
// bytes, what `n_shards` > 1 asks of msufsort_hip_make_sa_*_dev, and what the C-ABI multi-GPU entry runs per device.
// The shards take turns on the one workspace; every slice is written straight into the caller's array.
// ------------------------------------------------------------------------------------------------
template <bool W>
int build_logical(msufsort_hip_ctx* c, u8* d_text, u64 n, typename Wd<W>::sa_t* d_sa, int G, const msufsort_hip_opts* opts_in)
{
    typedef typename Wd<W>::sa_t sa_t;
    hipStream_t st = c->stream;
    const int verbose = opts_in ? opts_in->verbose : 0;
    u64 z = 0;
    TRY(trailing_zeros(c, d_text, n, &z));
    ShardCuts sc;
    TRY(plan_shards<W>(c, d_text, n, z, G, sc));
    TRY(c->grp_full.ensure((size_t)(n + 1) * 4));
    u32* grp = c->grp_full.as<u32>();
    msufsort_hip_opts o{};
    if (opts_in) o = *opts_in;
    o.n_shards = std::max(G, 2);          // (forces the "publish groups" exit; a single wide shard takes it anyway)
    if (o.text_rounds <= 0) o.text_rounds = W ? 3 : 8;
    {   // size the one workspace for the largest shard (growing it shard by shard would free and reallocate all of it)
        u64 mx = 0;
        for (int g = 0; g < G; ++g) mx = std::max(mx, sc.rows[g + 1] - sc.rows[g]);
        TRY(c->ensure_workspace(mx));
    }
    bool any = false;
    u64 depth = 0;
    msufsort_hip_timings acc{};
    auto t_start = std::chrono::steady_clock::now();
    for (int g = 0; g < G; ++g) {
        const u64 lo = sc.rows[g], hi = sc.rows[g + 1];
        if (hi == lo) continue;
        o.shard = g;
        c->hint_small_alphabet = c->plan_small_alphabet;
        const int r = build_sa<W>(c, d_text, n, d_sa + lo, lo, z, sc.cuts[g], sc.cuts[g + 1], sc.rank0[g], g == 0, &o, n - z > 0, grp + lo, hi - lo);
        if (r < 0) return r;
        acc.hist16_ms += c->tm.hist16_ms; acc.scatter0_ms += c->tm.scatter0_ms; acc.scatter1_ms += c->tm.scatter1_ms;
        acc.bucket_sort_ms += c->tm.bucket_sort_ms; acc.refine_ms += c->tm.refine_ms; acc.rounds = std::max(acc.rounds, c->tm.rounds);
        acc.unresolved_after_round0 += c->tm.unresolved_after_round0; acc.gathered_records += c->tm.gathered_records;
        if (r == MSUFSORT_HIP_UNRESOLVED) {
            const u64 d = (u64)c->tm.stop_depth;
            if (any && d != depth) { set_error("shards stopped at different depths (%llu, %llu)", (unsigned long long)depth, (unsigned long long)d); return MSUFSORT_HIP_ERR_INTERNAL; }
            any = true; depth = d;
        }
    }
    int steps = 0;
    double dbl_ms = 0;
    if (any) {
        auto t_d = std::chrono::steady_clock::now();
        TRY(c->grp_prev.ensure((size_t)(n + 1) * 4));
        TRY(c->isa.ensure((size_t)(n + 2) * sizeof(sa_t)));
        sa_t* isa = c->isa.as<sa_t>();
        u32* grp_prev = c->grp_prev.as<u32>();
        for (int g = 0; g < G; ++g) {
            const u64 lo = sc.rows[g], hi = sc.rows[g + 1];
            if (hi > lo) hipLaunchKernelGGL(k_isa_from_slice<W>, dim3(grid_for(hi - lo)), dim3(256), 0, st, d_sa + lo, grp + lo, hi - lo, lo, isa);
        }
        const u64 win = std::min<u64>(n + 1, 1ull << 27);          // rows per rank-update window
        TRY(c->upd.ensure((size_t)win * (W ? 16 : 8)));
        std::vector<char> live(G, 1);
        std::vector<u64> items(G, 0);
        for (auto& a : c->active) a.release();
        c->active.assign(G, ActiveSet());
        for (u64 h = depth;; h *= 2, ++steps) {
            u64 tied_total = 0;
            for (int g = 0; g < G; ++g) {
                const u64 lo = sc.rows[g], hi = sc.rows[g + 1];
                if (hi == lo || !live[g]) continue;
                u64 t = 0;
                TRY((double_sort<W>(c, c->active[g], n, d_sa + lo, grp + lo, grp_prev + lo, hi - lo, isa, h, verbose, &t, &items[g])));
                if (t == 0) live[g] = 0;           // (nothing tied: no updates either)
                else acc.doubling_records += (int64_t)items[g];
            }
            // the ranks are read-only while ANY shard still sorts with them: the updates of all shards are applied afterwards
            for (int g = 0; g < G; ++g) {
                const u64 lo = sc.rows[g], hi = sc.rows[g + 1];
                if (hi == lo || !live[g]) continue;
                for (u64 i0 = 0; i0 < items[g]; i0 += win) {
                    const u64 i1 = std::min(items[g], i0 + win);
                    u64 cnt = 0, tied = 0;
                    TRY((emit_updates<W>(c, c->active[g], d_sa + lo, grp + lo, grp_prev + lo, hi - lo, lo, i0, i1, items[g], c->upd.as<u64>(), win, &cnt, &tied)));
                    TRY((apply_updates<W>(c, c->upd.as<u64>(), cnt, isa)));
                    tied_total += tied;
                }
            }
            if (verbose) fprintf(stderr, "[msufsort_hip] doubling step %d (h = %llu): %llu rows still tied\n", steps, (unsigned long long)h, (unsigned long long)tied_total);
            if (tied_total == 0) { ++steps; break; }
            if (h > 2 * n + 2) { set_error("prefix doubling did not converge"); return MSUFSORT_HIP_ERR_INTERNAL; }
        }
        HIP_TRY(hipStreamSynchronize(st));
        for (auto& a : c->active) a.release();
        c->active.clear();
        dbl_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_d).count();
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    acc.n = (int64_t)n; acc.m = (int64_t)(n - z);
    acc.doubling_rounds = steps;
    acc.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    acc.stop_depth = (int64_t)depth;
    acc.logical_shards = G;
    acc.other_ms = dbl_ms;              // wall time of the distributed doubling phase
    c->tm = acc;
    return MSUFSORT_HIP_OK;
}

// number of logical shards a single-process build needs so that one shard's workspace (~70 B per suffix) fits
int auto_shards(msufsort_hip_ctx* c, u64 n, bool wide)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)64 << 30;
    free_b += c->held_bytes();                                            // what this context already holds is reused, not needed again
    const u64 fixed = (n + 1) * (wide ? 8 + 4 + 4 : 4 + 4 + 4) + n + ((u64)4 << 30);          // rank array + grp + grp_prev; lists of tied rows, update window
    const u64 avail = free_b > fixed + ((u64)8 << 30) ? free_b - fixed - ((u64)8 << 30) : ((u64)4 << 30);
    u64 per = std::min<u64>(avail / 80, wide ? (1ull << 30) : (1ull << 31));
    per = std::max<u64>(per, 1u << 20);
    return (int)std::max<u64>(1, (n + per - 1) / per);
}

int zero_pad(msufsort_hip_ctx* c, u8* d_text, u64 n)
{
    HIP_TRY(hipMemsetAsync(d_text + n, 0, MSUFSORT_HIP_TEXT_PAD, c->stream));
    return MSUFSORT_HIP_OK;
}

int check_n(int64_t n)
{
    if (n < 0) { set_error("negative length"); return MSUFSORT_HIP_ERR_BAD_ARG; }
    if (n > 0x7ffffffeLL) { set_error("n = %lld exceeds the int32 limit 2^31-2 (use the int64 entry points)", (long long)n); return MSUFSORT_HIP_ERR_TOO_LARGE; }
    return MSUFSORT_HIP_OK;
}

int check_n64(int64_t n)
{
    if (n < 0) { set_error("negative length"); return MSUFSORT_HIP_ERR_BAD_ARG; }
    if (n > (int64_t)((1ull << 40) - 2)) { set_error("n = %lld exceeds the 40-bit index limit", (long long)n); return MSUFSORT_HIP_ERR_TOO_LARGE; }
    return MSUFSORT_HIP_OK;
}

// Contexts of the host-pointer entry points (the one-shot functions and msufsort_hip_make_sa_multi): kept for the life of
// the process and handed out exclusively - creating one costs a stream and, on first use, the hipMalloc of the workspace
// (seconds for a 1 GiB input), which the reference pays once per msufsort instance for its worker pool (msufsort.h:311-388)
// and a one-shot call here must not pay every time.  msufsort_hip_release_cached() frees them.
struct CtxPool {
    struct Slot { int device; msufsort_hip_ctx* c; bool busy; };
    std::mutex mu;
    std::vector<Slot> slots;
    int acquire(int device, msufsort_hip_ctx** out)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto& s : slots) if (s.device == device && !s.busy) { s.busy = true; *out = s.c; return MSUFSORT_HIP_OK; }
        }
        msufsort_hip_ctx* c = nullptr;
        TRY(msufsort_hip_ctx_create(&c, device, 0));
        std::lock_guard<std::mutex> lk(mu);
        slots.push_back({device, c, true});
        *out = c;
        return MSUFSORT_HIP_OK;
    }
    void release(msufsort_hip_ctx* c)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& s : slots) if (s.c == c) s.busy = false;
    }
    void clear()
    {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<Slot> keep;
        for (auto& s : slots) { if (s.busy) keep.push_back(s); else msufsort_hip_ctx_destroy(s.c); }
        slots.swap(keep);
    }
};
CtxPool g_pool;

struct TmpCtx {
    msufsort_hip_ctx* c = nullptr;
    int acquire(int device) { return g_pool.acquire(device, &c); }
    ~TmpCtx() { if (c) g_pool.release(c); }
};

}  // namespace

#include "induce_host.inc"

// ================================================================================================
extern "C" {

int msufsort_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* msufsort_hip_strerror(int status)
{
    switch (status) {
        case MSUFSORT_HIP_OK: return "ok";
        case MSUFSORT_HIP_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case MSUFSORT_HIP_ERR_BAD_ARG: return "bad argument";
        case MSUFSORT_HIP_ERR_TOO_LARGE: return "input too large for this interface";
        case MSUFSORT_HIP_ERR_HIP: return "HIP runtime error";
        case MSUFSORT_HIP_ERR_NOMEM: return "out of device memory";
        case MSUFSORT_HIP_ERR_INTERNAL: return "internal error";
        case MSUFSORT_HIP_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}

const char* msufsort_hip_last_error(void) { return g_last_error.c_str(); }

#ifndef MSUFSORT_HIP_BUILD_ID
#define MSUFSORT_HIP_BUILD_ID "unknown"
#endif
const char* msufsort_hip_build_id(void) { return MSUFSORT_HIP_BUILD_ID; }

int msufsort_hip_ctx_create(msufsort_hip_ctx** out, int32_t device, int64_t max_n)
{
    if (!out) return MSUFSORT_HIP_ERR_BAD_ARG;
    *out = nullptr;
    if (msufsort_hip_device_count() <= 0) { set_error("no HIP device visible"); return MSUFSORT_HIP_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    auto* c = new msufsort_hip_ctx();
    c->device = device;
    c->sw.load();
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; set_error("hipStreamCreate: %s", hipGetErrorString(e)); return MSUFSORT_HIP_ERR_HIP; }
    for (auto& ev : c->ev) (void)hipEventCreate(&ev);
    (void)hipHostMalloc(reinterpret_cast<void**>(&c->h_counters), C_NCOUNTERS * 4, hipHostMallocDefault);
    (void)hipHostMalloc(reinterpret_cast<void**>(&c->h_hist), 65536 * 8, hipHostMallocDefault);
    (void)hipHostMalloc(reinterpret_cast<void**>(&c->h_upd), 64, hipHostMallocDefault);
    if (!c->h_counters || !c->h_hist || !c->h_upd) { msufsort_hip_ctx_destroy(c); set_error("hipHostMalloc failed"); return MSUFSORT_HIP_ERR_NOMEM; }
    if (max_n > 0) {
        int r = c->ensure_workspace((u64)max_n);
        if (r) { msufsort_hip_ctx_destroy(c); return r; }
    }
    *out = c;
    return MSUFSORT_HIP_OK;
}

void msufsort_hip_ctx_destroy(msufsort_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    delete c->ring; c->ring = nullptr;
    c->release_all();
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_hist) (void)hipHostFree(c->h_hist);
    if (c->h_upd) (void)hipHostFree(c->h_upd);
    if (c->h_ind) (void)hipHostFree(c->h_ind);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->gather_stream) (void)hipStreamDestroy(c->gather_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* msufsort_hip_ctx_stream(msufsort_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }

int msufsort_hip_ctx_sync(msufsort_hip_ctx* c)
{
    if (!c) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MSUFSORT_HIP_OK;
}

void msufsort_hip_release_cached(void) { g_pool.clear(); }

int msufsort_hip_ctx_trim(msufsort_hip_ctx* c)
{
    if (!c) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    {   // the pinned ring of the host-pointer entry points (8 x 32 MiB + its copy threads) goes with the workspace; it is rebuilt on demand
        std::lock_guard<std::mutex> lk(c->ring_mu);
        std::lock_guard<std::mutex> use(c->ring_use);
        delete c->ring; c->ring = nullptr;
    }
    c->release_all();
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_last_timings(msufsort_hip_ctx* c, msufsort_hip_timings* out)
{
    if (!c || !out) return MSUFSORT_HIP_ERR_BAD_ARG;
    *out = c->tm;
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_make_sa_i32_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int32_t* d_sa_out, const msufsort_hip_opts* opts)
{
    if (!c || !d_sa_out || (n > 0 && !d_text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n(n));
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) { HIP_TRY(hipMemsetAsync(d_sa_out, 0, 4, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); return MSUFSORT_HIP_OK; }
    TRY(zero_pad(c, d_text, (u64)n));
    if (opts && opts->n_shards > 1 && opts->shard < 0)      // shard = -1: all n_shards logical shards, one after the other, on this GPU
        return build_logical<false>(c, d_text, (u64)n, reinterpret_cast<u32*>(d_sa_out), opts->n_shards, opts);
    u64 z = 0;
    u32 tail_values = 0;
    TRY(trailing_zeros(c, d_text, (u64)n, &z, &tail_values));
    msufsort_hip_opts o{};
    if (opts) o = *opts;
    o.n_shards = 1; o.shard = 0;
    // two-stage build (B* sort + induction) for text-like inputs; everything it declines goes through the sort-all path.
    // The default policy first looks at what is on the host anyway (length; the byte values among the last 4 KiB, which
    // trailing_zeros fetched, and three 1 KiB samples of the body that came with them): inputs that cannot qualify - short ones,
    // random bytes - do not pay for the typing passes.
    c->sw.load();
    int two_stage = o.two_stage;
    if (c->sw.two_stage_set) two_stage = c->sw.two_stage;
    bool hist_done = false;
    int why = 0;                 // why a two-stage attempt was handed back (IND_WHY_*; 0: not tried / nothing spent)
    const bool few_values = tail_values <= 84u;          // (byte values among the tail and three body samples)
    c->hint_small_alphabet = few_values;
    if (two_stage > 0 || (two_stage == 0 && (u64)n >= ((u64)TWO_STAGE_MIN_MIB_TEXT << 20) && tail_values <= 128u)) {
        const int r = build_sa_two_stage(c, d_text, (u64)n, reinterpret_cast<u32*>(d_sa_out), z, &o, two_stage > 0, &hist_done, &why);
        if (r != MSUFSORT_HIP_UNRESOLVED) return r;
        c->hint_small_alphabet = few_values;
        // whatever left for the host already is rebuilt and sent again - AFTER the stale copies have landed: the ring's copy
        // threads do not finish in order, a late stale chunk must not overwrite a fresh one (round-4 advisor finding)
        if (c->sink && c->sink_rows) c->sink->flush();
        c->sink_rows = 0;
    }
    const int r = build_sa<false>(c, d_text, (u64)n, reinterpret_cast<u32*>(d_sa_out), 0, z, 0, 1ull << 32, z, true, &o, hist_done);
    if (why) c->tm.fallbacks = 1 | ((int64_t)why << 8);      // an abandoned two-stage attempt: its device time is part of this build
    if (hist_done && r == MSUFSORT_HIP_OK) {
        // the histogram (and whatever else the attempt ran) happened before build_sa started its clock: bill it
        float ms_ = 0;
        (void)hipEventElapsedTime(&ms_, c->ev[6], c->ev[8]); c->tm.hist16_ms = ms_;
        (void)hipEventElapsedTime(&ms_, c->ev[6], c->ev[5]); c->tm.total_ms = ms_;
    }
    return r;
}

int msufsort_hip_make_sa_two_stage_sharded_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int32_t* d_sa_out, uint32_t* d_bstar, int64_t bstar_capacity,
                                               msufsort_hip_exchange_fn exchange, void* user, const msufsort_hip_opts* opts)
{
    if (!c || !d_sa_out || !d_bstar || !opts || opts->n_shards < 1 || opts->shard < -1 || opts->shard >= opts->n_shards || bstar_capacity < 0 || (n > 0 && !d_text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n(n));
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) { HIP_TRY(hipMemsetAsync(d_sa_out, 0, 4, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); return MSUFSORT_HIP_OK; }
    TRY(zero_pad(c, d_text, (u64)n));
    u64 z = 0;
    TRY(trailing_zeros(c, d_text, (u64)n, &z));
    c->sw.load();
    // The contract: `exchange` runs ONCE on every rank - the ranks meet in its collective.  A rank that leaves the build before it
    // got there (an allocation that failed, a decline before the B* sort, the capacity check) would leave the healthy ranks waiting
    // in theirs (round-4 advisor finding), so the call is made on its behalf: status 1 = "declined" when the build handed the
    // input back (every rank alike), 2 = "this rank failed" - and every rank that is told 2 returns an error instead of going on
    // to collectives the failed rank will never join.
    struct Once { msufsort_hip_exchange_fn fn; void* user; bool called; int agreed; };
    Once once{exchange, user, false, 0};
    auto relay = [](void* u, const int64_t* bounds, int32_t ns, int32_t my_status) -> int {
        Once* o = static_cast<Once*>(u);
        o->called = true;
        o->agreed = o->fn(o->user, bounds, ns, my_status);
        return o->agreed;
    };
    TwoStageShards sh;
    sh.n_shards = opts->n_shards; sh.shard = opts->shard; sh.d_sstar = d_bstar; sh.sstar_capacity = (u64)bstar_capacity;
    sh.exchange = exchange ? +relay : nullptr; sh.user = &once;
    msufsort_hip_opts o = *opts;
    o.n_shards = 1; o.shard = 0;
    bool hist_done = false;
    int why = 0;
    int r = build_sa_two_stage(c, d_text, (u64)n, reinterpret_cast<u32*>(d_sa_out), z, &o, opts->two_stage > 0, &hist_done, &why, &sh);
    if (exchange && !once.called) {
        std::vector<int64_t> none((size_t)opts->n_shards + 1, 0);
        const std::string keep = g_last_error;
        (void)relay(&once, none.data(), opts->n_shards, r < 0 ? 2 : 1);
        if (r < 0) g_last_error = keep;
    }
    if (exchange && once.agreed >= 2 && r >= 0) { set_error("two-stage: a peer rank failed before the exchange of the sorted B* slices"); return MSUFSORT_HIP_ERR_INTERNAL; }
    if (r == MSUFSORT_HIP_UNRESOLVED) return why == IND_WHY_LOOKBACK ? MSUFSORT_HIP_TWO_STAGE_FAILED_LOCALLY : MSUFSORT_HIP_TWO_STAGE_DECLINED;
    return r;
}

int msufsort_hip_shard_bounds_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int32_t n_shards, int64_t* bounds)
{
    if (!c || !bounds || n_shards < 1 || (n > 0 && !d_text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n64(n));
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) { for (int g = 0; g <= n_shards; ++g) bounds[g] = g ? 1 : 0; return MSUFSORT_HIP_OK; }
    TRY(zero_pad(c, d_text, (u64)n));
    u64 z = 0;
    TRY(trailing_zeros(c, d_text, (u64)n, &z));
    ShardCuts sc;
    if (n > 0x7ffffffeLL) TRY((plan_shards<true>(c, d_text, (u64)n, z, n_shards, sc)));
    else TRY((plan_shards<false>(c, d_text, (u64)n, z, n_shards, sc)));
    for (int g = 0; g <= n_shards; ++g) bounds[g] = (int64_t)sc.rows[g];
    return MSUFSORT_HIP_OK;
}

// ---- the 16-bit histogram computed sharded (SURVEY 8(e) "Partitioning": "if computed sharded: one all-reduce") ----
int msufsort_hip_hist_part_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int32_t part, int32_t parts, uint64_t* d_hist_out, int32_t* stripes_out)
{
    if (!c || !d_text || !d_hist_out || n < 1 || parts < 1 || part < 0 || part >= parts) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n64(n));
    HIP_TRY(hipSetDevice(c->device));
    c->xh.reset();
    c->plan_cache.reset();
    TRY(zero_pad(c, d_text, (u64)n));
    u64 z = 0;
    TRY(trailing_zeros(c, d_text, (u64)n, &z));
    const u64 m = (u64)n - z;
    HIP_TRY(hipMemsetAsync(d_hist_out, 0, 65536 * 8, c->stream));
    u32 s0 = 0, s1 = 0, ns = 0;
    if (m > 0) {
        u32 hchunks = 0;
        TRY(plan_stripes(c, m, &hchunks));
        ns = c->nchunks;
        s0 = (u32)((u64)ns * (u32)part / (u32)parts); s1 = (u32)((u64)ns * ((u32)part + 1) / (u32)parts);
        // one workgroup per histogram chunk: a part of the stripes is cut into as many chunks as the whole text would be (>= 64 KiB each)
        u32 per = c->hist_per;
        while ((u64)(s1 - s0) * per < 256 && c->chunk_len / (per * 2) >= 65536 && (c->chunk_len % (per * 2 * 16)) == 0) per *= 2;
        c->xh.per = per;
        const u32 c0 = s0 * per, c1 = s1 * per;
        if (c1 > c0) {
            TRY(c->hist_partial.ensure((size_t)std::max<u32>(c1 - c0, 256u) * 65536 * 4));
            hipLaunchKernelGGL(k_hist16<0>, dim3(c1 - c0), dim3(1024), H16_LDS_BYTES, c->stream, d_text, m, (u32)(c->chunk_len / per), c1, c->hist_partial.as<u32>(), 0u,
                               (const unsigned short*)nullptr, c0);
            hipLaunchKernelGGL(k_reduce16_part, dim3(256), dim3(256), 0, c->stream, c->hist_partial.as<u32>(), c1 - c0, reinterpret_cast<u64*>(d_hist_out));
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    if (stripes_out) { stripes_out[0] = (int32_t)ns; stripes_out[1] = (int32_t)s0; stripes_out[2] = (int32_t)s1; }
    c->xh.stage = 1; c->xh.text = d_text; c->xh.n = (u64)n; c->xh.z = z; c->xh.s0 = s0; c->xh.s1 = s1;
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_hist_plan_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int32_t n_shards, const uint64_t* d_hist_sum,
                               uint32_t* d_sums_out, int32_t stripes_per_part, int64_t* bounds_out)
{
    if (!c || !d_text || !d_hist_sum || !d_sums_out || !bounds_out || n_shards < 1 || stripes_per_part < 0) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    auto& x = c->xh;
    if (x.stage != 1 || x.text != d_text || x.n != (u64)n) { x.reset(); set_error("hist_plan: no histogram part of this text in the context (call msufsort_hip_hist_part_dev first)"); return MSUFSORT_HIP_ERR_BAD_ARG; }
    if ((int32_t)(x.s1 - x.s0) > stripes_per_part) { x.reset(); set_error("hist_plan: %u stripes in my part, room for %d", x.s1 - x.s0, stripes_per_part); return MSUFSORT_HIP_ERR_BAD_ARG; }
    const u64 z = x.z, m = (u64)n - z;
    c->sw.load();
    TRY(c->xh_hist.ensure(65536 * 8));
    HIP_TRY(hipMemcpyAsync(c->h_hist, d_hist_sum, 65536 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->xh_hist.p, d_hist_sum, 65536 * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_sums_out, 0, (size_t)n_shards * (size_t)stripes_per_part * 256 * 4, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<u64> bs(65537);
    bs[0] = 0;
    for (u32 k = 0; k < 65536; ++k) bs[k + 1] = bs[k] + c->h_hist[k];
    if (bs[65536] != m) { x.reset(); set_error("hist_plan: the summed histogram counts %llu suffixes, the text has %llu", (unsigned long long)bs[65536], (unsigned long long)m); return MSUFSORT_HIP_ERR_BAD_ARG; }
    {
        u32 nv = 0;
        for (u32 b = 0; b < 256; ++b) nv += bs[(b + 1) * 256] != bs[b * 256];
        x.small_alphabet = nv <= 83;
    }
    // the plan of plan_shards, two-byte keys only: a cut that would be refined inside a heavy key (DNA, text) needs the
    // per-stripe counts of that key's deeper histogram from every rank - those inputs keep the replicated histogram
    x.cuts.assign(n_shards + 1, 0); x.rows.assign(n_shards + 1, 0); x.rank0.assign(n_shards + 1, z);
    x.cuts[n_shards] = 1ull << 32; x.rows[n_shards] = (u64)n + 1;
    const u64 tol = std::max<u64>(m / ((u64)n_shards * 16), 1);
    for (int g = 1; g < n_shards; ++g) {
        const u64 target = (u64)((unsigned __int128)m * (u64)g / (u64)n_shards);
        const u32 k = (u32)(std::lower_bound(bs.begin(), bs.begin() + 65536, target) - bs.begin());
        if (!c->sw.no_refine && k > 0 && bs[k] - target > tol) { x.reset(); return MSUFSORT_HIP_HIST_NEEDS_REPLICA; }
        u64 cut = (u64)k << 16, before = bs[k];
        if (cut < x.cuts[g - 1]) { cut = x.cuts[g - 1]; before = x.rank0[g - 1] - z; }
        x.cuts[g] = cut; x.rows[g] = 1 + z + before; x.rank0[g] = z + before;
    }
    x.rank0[n_shards] = z + m;
    // what every shard's scatter needs from my stripes: first-byte sums over ITS key range
    if (x.s1 > x.s0)
        for (int g0 = 0; g0 < n_shards; g0 += 64) {          // (all cuts are two-byte key boundaries here; one pass over my partials per 64 shards)
            ShardKeys keys;
            keys.n = (u32)std::min(64, n_shards - g0); keys.first = (u32)g0;
            for (u32 i = 0; i <= keys.n; ++i) keys.k[i] = (u32)(x.cuts[g0 + i] >> 16);
            hipLaunchKernelGGL(k_stripe_sums_multi, dim3((x.s1 - x.s0) * 16), dim3(256), 0, c->stream, c->hist_partial.as<u32>(), x.per, keys, d_sums_out, (u32)stripes_per_part);
        }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    for (int g = 0; g <= n_shards; ++g) bounds_out[g] = (int64_t)x.rows[g];
    x.n_shards = n_shards; x.stage = 2;
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_hist_install_dev(msufsort_hip_ctx* c, int32_t shard, const uint32_t* d_stripe_sums, int32_t stripes)
{
    if (!c || !d_stripe_sums) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    auto& x = c->xh;
    if (x.stage != 2 || shard < 0 || shard >= x.n_shards || (u32)stripes != c->nchunks) { x.reset(); set_error("hist_install: no plan in the context, or the stripe count is not the plan's"); return MSUFSORT_HIP_ERR_BAD_ARG; }
    TRY(c->xh_sums.ensure(128 * 256 * 4));
    HIP_TRY(hipMemcpyAsync(c->xh_sums.p, d_stripe_sums, (size_t)stripes * 256 * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    x.shard = shard; x.stage = 3;
    return MSUFSORT_HIP_OK;
}

}  // extern "C"

namespace {
template <bool W>
int make_sa_shard_impl(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, typename Wd<W>::sa_t* d_slice_out, uint32_t* d_grp_slice_out,
                       int64_t slice_capacity, int64_t* slice_lo, int64_t* slice_hi, const msufsort_hip_opts* opts)
{
    if (!c || !d_slice_out || !opts || opts->n_shards < 1 || opts->shard < 0 || opts->shard >= opts->n_shards || (n > 0 && !d_text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(W ? check_n64(n) : check_n(n));
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) {
        if (slice_lo) *slice_lo = 0;
        if (slice_hi) *slice_hi = opts->shard == 0 ? 1 : 0;
        if (opts->shard == 0) {
            HIP_TRY(hipMemsetAsync(d_slice_out, 0, sizeof(typename Wd<W>::sa_t), c->stream));
            if (d_grp_slice_out) HIP_TRY(hipMemsetAsync(d_grp_slice_out, 0, 4, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        return MSUFSORT_HIP_OK;
    }
    const bool planned = c->xh.stage == 3 && c->xh.text == d_text && c->xh.n == (u64)n && c->xh.n_shards == opts->n_shards && c->xh.shard == opts->shard;
    u64 z = 0;
    if (planned) z = c->xh.z;      // (hist_part padded the text and counted its trailing zero bytes)
    else {
        TRY(zero_pad(c, d_text, (u64)n));
        TRY(trailing_zeros(c, d_text, (u64)n, &z));
    }
    const u64 m = (u64)n - z;
    ShardCuts sc;
    bool ext_sums = false;
    HIP_TRY(hipEventRecord(c->ev[8], c->stream));
    auto& pcache = c->plan_cache;
    const bool cached = !planned && opts->reuse_plan && pcache.valid && pcache.wide == W && pcache.text == d_text && pcache.n == (u64)n && pcache.z == z &&
                        pcache.n_shards == opts->n_shards && m > 0;
    if (planned && m > 0) {
        // the histogram was computed sharded and all-reduced, the plan is made, my shard's stripe sums are here: no pass over the text
        sc.cuts = c->xh.cuts; sc.rows = c->xh.rows; sc.rank0 = c->xh.rank0;
        hipLaunchKernelGGL(k_hist_from_u64<W>, dim3(256), dim3(256), 0, c->stream, c->xh_hist.as<u64>(), c->hist.as<typename Wd<W>::hist_t>());
        c->plan_small_alphabet = c->xh.small_alphabet;
        ext_sums = true;
        c->sub_invalidate();
        // the plan stays (stage 2): the stripe sums of ANOTHER shard of this text may be installed next (a rank that sorts its key range
        // as several sub-shards, dist.py); any other call drops it
        c->xh.stage = 2; c->xh.shard = -1;
        pcache.reset();
    } else if (cached) {
        // another shard of the text the previous shard call planned: its histogram (c->hist, c->hist_partial) and cuts serve this one
        c->xh.reset();
        sc.cuts = pcache.cuts; sc.rows = pcache.rows; sc.rank0 = pcache.rank0;
        c->plan_small_alphabet = pcache.small_alphabet;
    } else {
        c->xh.reset();
        pcache.reset();
        TRY(plan_shards<W>(c, d_text, (u64)n, z, opts->n_shards, sc));
        pcache.valid = true; pcache.wide = W; pcache.text = d_text; pcache.n = (u64)n; pcache.z = z; pcache.n_shards = opts->n_shards;
        pcache.small_alphabet = c->plan_small_alphabet; pcache.cuts = sc.cuts; pcache.rows = sc.rows; pcache.rank0 = sc.rank0;
    }
    HIP_TRY(hipEventRecord(c->ev[9], c->stream));
    const int g = opts->shard;
    const u64 lo = sc.rows[g], hi = sc.rows[g + 1];
    if (slice_lo) *slice_lo = (int64_t)lo;
    if (slice_hi) *slice_hi = (int64_t)hi;
    if ((int64_t)(hi - lo) > slice_capacity) { set_error("slice needs %llu rows, capacity %lld", (unsigned long long)(hi - lo), (long long)slice_capacity); return MSUFSORT_HIP_ERR_BAD_ARG; }
    if (hi == lo) return MSUFSORT_HIP_OK;
    c->hint_small_alphabet = c->plan_small_alphabet;
    c->ext_stripe_sums = ext_sums;
    const int r = build_sa<W>(c, d_text, (u64)n, d_slice_out, lo, z, sc.cuts[g], sc.cuts[g + 1], sc.rank0[g], g == 0, opts, m > 0, d_grp_slice_out, hi - lo);
    c->ext_stripe_sums = false;
    if (r == MSUFSORT_HIP_OK || r == MSUFSORT_HIP_UNRESOLVED) {
        // the histogram (every shard reads the whole text) and the planning of the cuts ran before build_sa started its clock
        float ms_ = 0;
        (void)hipEventElapsedTime(&ms_, c->ev[8], c->ev[9]);
        c->tm.hist16_ms += ms_; c->tm.total_ms += ms_;
    }
    return r;
}

__global__ __launch_bounds__(256) void k_widen(const int32_t* __restrict__ in, u64 rows, int64_t* __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < rows; i += (u64)gridDim.x * 256u) out[i] = (int64_t)in[i];
}
}  // namespace

extern "C" {

int msufsort_hip_make_sa_shard_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int32_t* d_slice_out, int64_t slice_capacity,
                                   int64_t* slice_lo, int64_t* slice_hi, const msufsort_hip_opts* opts)
{
    return make_sa_shard_impl<false>(c, d_text, n, reinterpret_cast<u32*>(d_slice_out), nullptr, slice_capacity, slice_lo, slice_hi, opts);
}

int msufsort_hip_make_sa_shard_groups_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int32_t* d_slice_out, uint32_t* d_grp_slice_out,
                                          int64_t slice_capacity, int64_t* slice_lo, int64_t* slice_hi, int64_t* depth_out,
                                          const msufsort_hip_opts* opts)
{
    if (!d_grp_slice_out) return MSUFSORT_HIP_ERR_BAD_ARG;
    const int r = make_sa_shard_impl<false>(c, d_text, n, reinterpret_cast<u32*>(d_slice_out), d_grp_slice_out, slice_capacity, slice_lo, slice_hi, opts);
    if (depth_out) *depth_out = (r == MSUFSORT_HIP_UNRESOLVED) ? c->tm.stop_depth : 0;
    return r;
}

int msufsort_hip_make_sa_shard_groups_i64_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int64_t* d_slice_out, uint32_t* d_grp_slice_out,
                                              int64_t slice_capacity, int64_t* slice_lo, int64_t* slice_hi, int64_t* depth_out,
                                              const msufsort_hip_opts* opts)
{
    if (!d_grp_slice_out) return MSUFSORT_HIP_ERR_BAD_ARG;
    const int r = make_sa_shard_impl<true>(c, d_text, n, reinterpret_cast<u64*>(d_slice_out), d_grp_slice_out, slice_capacity, slice_lo, slice_hi, opts);
    if (depth_out) *depth_out = (r == MSUFSORT_HIP_UNRESOLVED) ? c->tm.stop_depth : 0;
    return r;
}

// ---- distributed prefix doubling: the per-shard pieces (msufsort_amd/dist.py and the single-process drivers use them) ----
int msufsort_hip_isa_from_slice_dev(msufsort_hip_ctx* c, const void* d_sa_slice, const uint32_t* d_grp_slice, int64_t lo, int64_t hi,
                                    void* d_isa, int32_t index_bytes)
{
    if (!c || !d_sa_slice || !d_grp_slice || !d_isa || lo < 0 || hi < lo || (index_bytes != 4 && index_bytes != 8)) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (hi > lo) {
        if (index_bytes == 8) hipLaunchKernelGGL(k_isa_from_slice<true>, dim3(grid_for((u64)(hi - lo))), dim3(256), 0, c->stream, static_cast<const u64*>(d_sa_slice), d_grp_slice, (u64)(hi - lo), (u64)lo, static_cast<u64*>(d_isa));
        else hipLaunchKernelGGL(k_isa_from_slice<false>, dim3(grid_for((u64)(hi - lo))), dim3(256), 0, c->stream, static_cast<const u32*>(d_sa_slice), d_grp_slice, (u64)(hi - lo), (u64)lo, static_cast<u32*>(d_isa));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_double_sort_dev(msufsort_hip_ctx* c, int64_t n, void* d_sa_slice, uint32_t* d_grp_slice, uint32_t* d_grp_prev_slice,
                                 int64_t lo, int64_t hi, const void* d_isa, int64_t h, int32_t index_bytes, const msufsort_hip_opts* opts,
                                 int64_t* tied_before, int64_t* emit_items)
{
    if (!c || !d_sa_slice || !d_grp_slice || !d_grp_prev_slice || !d_isa || n <= 0 || lo < 0 || hi < lo || h <= 0 || (index_bytes != 4 && index_bytes != 8)) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    u64 t = 0, items = 0;
    const int verbose = opts ? opts->verbose : 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (c->active.empty()) c->active.resize(1);
    if (hi > lo) {
        if (index_bytes == 8) TRY((double_sort<true>(c, c->active[0], (u64)n, static_cast<u64*>(d_sa_slice), d_grp_slice, d_grp_prev_slice, (u64)(hi - lo), static_cast<const u64*>(d_isa), (u64)h, verbose, &t, &items)));
        else TRY((double_sort<false>(c, c->active[0], (u64)n, static_cast<u32*>(d_sa_slice), d_grp_slice, d_grp_prev_slice, (u64)(hi - lo), static_cast<const u32*>(d_isa), (u64)h, verbose, &t, &items)));
    }
    c->tm.refine_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();     // this shard's sort work of the step
    if (tied_before) *tied_before = (int64_t)t;
    if (emit_items) *emit_items = (int64_t)items;
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_emit_updates_dev(msufsort_hip_ctx* c, const void* d_sa_slice, const uint32_t* d_grp_slice, const uint32_t* d_grp_prev_slice,
                                  int64_t lo, int64_t hi, int64_t i0, int64_t i1, int64_t items_total, void* d_updates, int64_t capacity, int32_t index_bytes,
                                  int64_t* count, int64_t* tied_rows)
{
    if (!c || !d_sa_slice || !d_grp_slice || !d_grp_prev_slice || !d_updates || lo < 0 || hi < lo || i0 < 0 || i1 < i0 || i1 > items_total || items_total > hi - lo || capacity < 0 ||
        (index_bytes != 4 && index_bytes != 8) || c->active.empty()) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    u64 cnt = 0, tied = 0;
    if (index_bytes == 8) TRY((emit_updates<true>(c, c->active[0], static_cast<const u64*>(d_sa_slice), d_grp_slice, d_grp_prev_slice, (u64)(hi - lo), (u64)lo, (u64)i0, (u64)i1, (u64)items_total, static_cast<u64*>(d_updates), (u64)capacity, &cnt, &tied)));
    else TRY((emit_updates<false>(c, c->active[0], static_cast<const u32*>(d_sa_slice), d_grp_slice, d_grp_prev_slice, (u64)(hi - lo), (u64)lo, (u64)i0, (u64)i1, (u64)items_total, static_cast<u64*>(d_updates), (u64)capacity, &cnt, &tied)));
    if (count) *count = (int64_t)cnt;
    if (tied_rows) *tied_rows = (int64_t)tied;
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_apply_updates_dev(msufsort_hip_ctx* c, const void* d_updates, int64_t count, void* d_isa, int32_t index_bytes)
{
    if (!c || count < 0 || (count > 0 && (!d_updates || !d_isa)) || (index_bytes != 4 && index_bytes != 8)) return MSUFSORT_HIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (index_bytes == 8) TRY((apply_updates<true>(c, static_cast<const u64*>(d_updates), (u64)count, static_cast<u64*>(d_isa))));
    else TRY((apply_updates<false>(c, static_cast<const u64*>(d_updates), (u64)count, static_cast<u32*>(d_isa))));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_make_sa_i32_ctx(msufsort_hip_ctx* c, const uint8_t* text, int64_t n, int32_t* sa_out, const msufsort_hip_opts* opts)
{
    if (!c || !sa_out || (n > 0 && !text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n(n));
    if (n == 0) { sa_out[0] = 0; return MSUFSORT_HIP_OK; }
    HIP_TRY(hipSetDevice(c->device));
    c->sw.load();
    HostTrace tr(c->sw.host_trace, "make_sa_i32_ctx");
    Prefault pf;
    TRY(c->text_own.ensure((size_t)n + MSUFSORT_HIP_TEXT_PAD));
    TRY(c->sa_own.ensure(((size_t)n + 1) * 4));
    TRY(copy_in(c, c->text_own.p, text, (size_t)n));
    tr.mark("H2D done");
    pf.start(sa_out, ((size_t)n + 1) * 4);           // the result is usually fresh memory: first touch while the text is sorted
    // text-like inputs (two-stage build): the rows of a bucket are final as soon as the last pass has left it - they travel while
    // the later buckets are still being induced (the pass takes ~13 ms of a 1 GiB text's 75: that much of the 76 ms D2H is hidden)
    const bool stream_rows = ((size_t)n + 1) * 4 >= ((size_t)64 << 20) && !c->sw.no_ring;
    Copier copier;
    if (stream_rows) {
        if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        copier.tr = &tr;
        copier.start(c->device, c->copy_stream, c);
        c->sink = &copier; c->sink_host = sa_out; c->sink_rows = 0;
    }
    c->trace = &tr;
    int r = msufsort_hip_make_sa_i32_dev(c, c->text_own.as<u8>(), n, c->sa_own.as<int32_t>(), opts);
    c->sink = nullptr; c->trace = nullptr;
    tr.mark("built");
    if (stream_rows) {
        if (r == MSUFSORT_HIP_OK) {
            if (c->sink_rows == (u64)n) copier.push({sa_out, c->sa_own.p, 4, nullptr});                          // rows 1 .. n are on their way: row 0
            else copier.push({sa_out, c->sa_own.p, ((size_t)n + 1) * 4, nullptr});                                // nothing (valid) left early: everything
        }
        copier.finish();
        if (r == MSUFSORT_HIP_OK && copier.status) { set_error("D2H of the rows failed"); r = copier.status; }
    } else if (r == MSUFSORT_HIP_OK) r = copy_out(c, c->stream, sa_out, c->sa_own.p, ((size_t)n + 1) * 4);
    tr.mark("copied out");
    pf.wait();
    tr.mark("first touch joined");
    return r;
}

int msufsort_hip_make_sa_i32(const uint8_t* text, int64_t n, int32_t* sa_out, const msufsort_hip_opts* opts)
{
    if (!sa_out || (n > 0 && !text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n(n));
    if (n == 0) { sa_out[0] = 0; return MSUFSORT_HIP_OK; }
    TmpCtx t;
    TRY(t.acquire(opts ? opts->device : 0));
    return msufsort_hip_make_sa_i32_ctx(t.c, text, n, sa_out, opts);
}

// ---- 64-bit output.  n <= 2^31 - 2: the narrow build, widened on the device (unless opts->force_wide).  Larger inputs
// (and force_wide): wide records (40-bit indices), logical shards that take turns on this GPU, distributed doubling. ----
int msufsort_hip_make_sa_i64_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, int64_t* d_sa_out, const msufsort_hip_opts* opts)
{
    if (!c || !d_sa_out || n < 0 || (n > 0 && !d_text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n64(n));
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) { HIP_TRY(hipMemsetAsync(d_sa_out, 0, 8, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); return MSUFSORT_HIP_OK; }
    const bool wide = n > 0x7ffffffeLL || (opts && opts->force_wide);
    if (!wide) {
        TRY(c->sa_own.ensure(((size_t)n + 1) * 4));
        TRY(msufsort_hip_make_sa_i32_dev(c, d_text, n, c->sa_own.as<int32_t>(), opts));
        hipLaunchKernelGGL(k_widen, dim3(std::min<u32>(cdiv((u64)n + 1, 256), 1u << 20)), dim3(256), 0, c->stream, c->sa_own.as<int32_t>(), (u64)n + 1, d_sa_out);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipGetLastError());
        return MSUFSORT_HIP_OK;
    }
    TRY(zero_pad(c, d_text, (u64)n));
    int G = auto_shards(c, (u64)n, true);
    if (opts && opts->n_shards > G) G = opts->n_shards;
    return build_logical<true>(c, d_text, (u64)n, reinterpret_cast<u64*>(d_sa_out), G, opts);
}

int msufsort_hip_make_sa_i64_ctx(msufsort_hip_ctx* c, const uint8_t* text, int64_t n, int64_t* sa_out, const msufsort_hip_opts* opts)
{
    if (!c || !sa_out || (n > 0 && !text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n64(n));
    if (n == 0) { sa_out[0] = 0; return MSUFSORT_HIP_OK; }
    HIP_TRY(hipSetDevice(c->device));
    Prefault pf;
    TRY(c->text_own.ensure((size_t)n + MSUFSORT_HIP_TEXT_PAD));
    TRY(c->aux2.ensure(((size_t)n + 1) * 8));
    TRY(copy_in(c, c->text_own.p, text, (size_t)n));
    pf.start(sa_out, ((size_t)n + 1) * 8);
    TRY(msufsort_hip_make_sa_i64_dev(c, c->text_own.as<u8>(), n, c->aux2.as<int64_t>(), opts));
    return copy_out(c, c->stream, sa_out, c->aux2.p, ((size_t)n + 1) * 8);
}

int msufsort_hip_make_sa_i64(const uint8_t* text, int64_t n, int64_t* sa_out, const msufsort_hip_opts* opts)
{
    if (!sa_out || (n > 0 && !text)) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n64(n));
    if (n == 0) { sa_out[0] = 0; return MSUFSORT_HIP_OK; }
    TmpCtx t;
    TRY(t.acquire(opts ? opts->device : 0));
    return msufsort_hip_make_sa_i64_ctx(t.c, text, n, sa_out, opts);
}

// Host-only: balanced key-range cuts from the exclusive prefix bstart[65537] of the 16-bit histogram (see plan_cuts64).
int msufsort_hip_plan_cuts(const uint64_t* bstart, int64_t n, int64_t z, int32_t n_shards, uint32_t* cuts, int64_t* rows)
{
    if (!bstart || !cuts || !rows || n_shards < 1 || n < 0 || z < 0 || z > n) return MSUFSORT_HIP_ERR_BAD_ARG;
    std::vector<u64> r(n_shards + 1);
    plan_cuts64(bstart, (u64)n, (u64)z, n_shards, cuts, r.data());
    for (int g = 0; g <= n_shards; ++g) rows[g] = (int64_t)r[g];
    return MSUFSORT_HIP_OK;
}

int msufsort_hip_debug_hist16_dev(msufsort_hip_ctx* c, uint8_t* d_text, int64_t n, uint32_t* d_hist)
{
    if (!c || !d_text || !d_hist || n <= 0) return MSUFSORT_HIP_ERR_BAD_ARG;
    TRY(check_n(n));
    HIP_TRY(hipSetDevice(c->device));
    TRY(zero_pad(c, d_text, (u64)n));
    TRY(run_hist<false>(c, d_text, (u64)n));
    HIP_TRY(hipMemcpyAsync(d_hist, c->hist.p, 65536 * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MSUFSORT_HIP_OK;
}

#include "bwt_host.inc"

}  // extern "C"

#include "multi_host.inc"
