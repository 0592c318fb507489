// Device kernels of the MI355X (gfx950) suffix-array builder.  Included once by engine.hip.
//
// What is being replaced: the stage-1 hot path of the reference ITS sort -
//   count_suffixes              reference msufsort.cpp:1496-1521  -> k_hist16 (+ k_reduce16/k_scan16)
//   bucket offsets              reference msufsort.cpp:1603-1630  -> k_scan16
//   initial_two_byte_radix_sort reference msufsort.cpp:1525-1555  -> k_scatter0 + k_partition (2 x 8 bit)
//   multikey_quicksort          reference msufsort.cpp:488-642    -> k_sort_fast2 / k_sort_mid / k_sort_tiny (LDS sorts on
//                                                                    big-endian key words) + k_partition levels
//   tandem-repeat shortcut      reference msufsort.cpp:316-484    -> prefix doubling on ranks: in place (k_isa_*, k_refill rank
//                                                                    mode) or distributed over shards (k_import_groups,
//                                                                    k_refill_rows, k_emit_updates, k_apply_updates)
// and, because every suffix is sorted here (no A/B/B* reduction), the stage-2 induction sweeps
// (cpp:646-1057) have no counterpart: the output of the sorts is already the final suffix array.
//
// Order semantics.  The reference orders unsigned bytes with "a proper prefix sorts first"
// (compare_suffixes cpp:147-180).  Here the text is treated as padded with infinitely many 0x00 bytes;
// two suffixes then compare equal forever only if both lie inside the trailing 0x00 run of the text
// (see DESIGN.md), so the host places those z suffixes first (ranks 0..z-1, descending index) and the
// kernels sort the remaining m = n - z suffixes with plain unsigned key comparisons and no tie rule.
//
// Records are 64-bit: (key32 << 32) | suffix_index.  key32 is a big-endian window of the text
// (get_value, cpp:129-143) or, in prefix-doubling rounds, the rank of suffix index+h.
//
// Index width (template parameter W of the kernels that look inside a record):
//   narrow (W = false): (key32 << 32) | index32; suffix-array rows and ranks are u32.  Inputs up to 2^31 - 2 bytes
//                       (the reference's own suffix_index is int32, msufsort.h:47, with two flag bits, h:84-93).
//   wide   (W = true):  (key24 << 40) | index40; rows and ranks are u64 (int64 output).  The upper word of a record is
//                       key24 << 8 | index bits 32..39, so everything that treats that word as "the key" still sorts
//                       correctly (the low byte only orders equal keys arbitrarily) - provided tie tests and LSD passes
//                       ignore its low KLOW = 8 bits.  Partition levels use key bytes at the same bit positions in both
//                       layouts (bits 56.., 48.., 40..); the fourth byte (bits 32..39) exists only in narrow records.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

template <bool W> struct Wd { typedef u32 sa_t; typedef u32 hist_t; };
template <> struct Wd<true> { typedef u64 sa_t; typedef u64 hist_t; };
template <bool W> __host__ __device__ constexpr u32 klow() { return W ? 8u : 0u; }
// full suffix index from the two words a sort kernel carries per record
template <bool W> __device__ __forceinline__ typename Wd<W>::sa_t full_idx(u32 keyword, u32 lo)
{
    if constexpr (W) return ((u64)(keyword & 255u) << 32) | (u64)lo; else return lo;
}
template <bool W> __device__ __forceinline__ typename Wd<W>::sa_t rec_idx(u64 r)
{
    if constexpr (W) return r & 0xffffffffffull; else return (u32)r;
}
// key = left-aligned 32-bit key word (wide: only its top 24 bits are kept)
template <bool W> __device__ __forceinline__ u64 make_rec(u32 key, typename Wd<W>::sa_t idx)
{
    if constexpr (W) return ((u64)(key & 0xffffff00u) << 32) | (u64)idx; else return ((u64)key << 32) | (u64)idx;
}

struct Desc {            // one segment = run of records that are equal on everything consumed so far
    u32 rec_off;         // first record (in record buffer `buf`)
    u32 len;
    u32 sa_off;          // first (shard-local) suffix-array row of the segment
    u32 buf;             // bits 0-1: record buffer index 0..2; bit 2: DESC_STALE; bits 8-15: kbits = number of low key bits that
                         // can still differ inside the segment (32 for fresh keys, 24 after the level-1 split, ...)
};
#define DESC_BUF(kbits, buf) (((u32)(kbits) << 8) | (u32)(buf))
// Prefix-doubling rounds keep ISA[i] = 1 + first row of i's group.  A segment that was emitted by the previous round's
// sort already has that rank in all its members, so after sorting it only the members that do NOT stay in its first
// run need a (random 4-byte) ISA write.  Segments created by a partition level in the current round carry DESC_STALE:
// their members still hold the parent's rank and are all rewritten.
#define DESC_STALE 4u

// x[] (optional, narrow builds of small alphabets): one 32-bit word per record of p[], same indexing - the key of the FIRST gather
// round, which k_scatter0 reads off the text tile it holds anyway; it travels with the records through round 0 and the round-0
// sorts put it into the free upper half of the records they emit, so that round 1 needs no gather
struct RecBufs { u64* p[3]; u32* x[3]; };

// counters block (device resident, read back by the host once per phase)
enum {
    C_POOL0 = 0, C_POOL1 = 1,         // tiny-pool element counts (slot 0/1)
    C_SEG0 = 2, C_SEG1 = 3,           // segment-array element counts
    C_LIST0 = 4,                      // [4..7]  slot 0: class A, B, C, large
    C_LIST1 = 8,                      // [8..11] slot 1
    C_LTILES0 = 12, C_LTILES1 = 13,   // tiles of the large list of slot 0/1
    C_LVL0 = 14, C_LVL1 = 15,         // level ping-pong large lists
    C_LVLT0 = 16, C_LVLT1 = 17,       // their tile counts
    C_ERR = 18,
    C_SENT = 19,                      // BWT sentinel row
    C_MS = 20,                        // suffixes in this shard
    C_RANK0 = 21,                     // global rank of the shard's first row
    C_FBB = 22, C_FBC = 23,           // segments k_sort_fast handed back (class B / C)
    C_HMAX = 24,                      // largest 16-bit bucket of this shard (skew forecast)
    C_HNZ = 25,                       // number of non-empty 16-bit buckets (alphabet-size estimate)
    C_ABITS = 26,                     // bits per symbol of the dense alphabet code (k_alphabet)
    C_ASIGMA = 27,                    // number of codes (symbols that occur + the reserved zero)
    C_CARRY = 28,                     // records k_carry_alloc placed in NEXT round's segment array during this round (a repeated
                                      // sort attempt restarts that array behind them, not at 0)
    C_CHAIN = 29,                     // suffixes k_chain_resolve finished (tie groups that are one arithmetic progression of positions)
    C_H17FLAG = 30,                   // k_hist17 / k_scan17: bit 0 an 8-bit LDS counter wrapped, bit 1 the 17-bit counts disagree with the 16-bit ones
    C_H17MAX = 31,                    // largest 17-bit bucket
    C_NCOUNTERS = 32
};

// size classes of the LDS sorts
#ifndef TINY_MAX
#define TINY_MAX 32
#endif
#define CLS_A_THREADS 64
#ifndef CLS_A_ITEMS
#define CLS_A_ITEMS 8        // <= 512 (12 / 16 measured on text: see profiles/HISTORY.md, tried and rejected)
#endif
#define CLS_B_THREADS 256
#define CLS_B_ITEMS 18       // <= 4608
#define CLS_C_THREADS 1024
#define CLS_C_ITEMS 18       // <= 18432
#define CAP_A (CLS_A_THREADS * CLS_A_ITEMS)
#define CAP_B (CLS_B_THREADS * CLS_B_ITEMS)
#define CAP_C (CLS_C_THREADS * CLS_C_ITEMS)

#ifndef P1_THREADS
#define P1_THREADS 1024
#endif
#ifndef P1_ITEMS
#define P1_ITEMS 8
#endif
#define P1_TILE (P1_THREADS * P1_ITEMS)      // 8192 records per tile
#define CUR0_STRIDE 64                       // u32 stride of the 256 first-byte cursors: one 256-B line each,
                                             // so the per-tile claim atomics spread over memory channels
#ifndef S0_POS
#define S0_POS 16                            // level-0 scatter: text positions per thread (8 or 16)
#endif
#ifndef S0_THREADS
#define S0_THREADS 1024
#endif
#define S0_TILE (S0_THREADS * S0_POS)        // text positions per workgroup

#define MODE_TEXT 0
#define MODE_ISA 1          // prefix doubling, ranks updated in place by the sorts (narrow single-GPU builds)
#define MODE_DEFER 2        // prefix doubling of a sharded / wide build: the rank array is read-only during a step; the sorts
                            // write grp_out[row] = first row of the row's tie group (coalesced, next to the suffix-array
                            // rows) and the rank updates are derived from it afterwards (k_emit_updates / k_apply_updates)

struct Lists {            // destination lists for segments discovered by a kernel
    Desc* cls[3];         // A, B, C
    u32 cap[3];
    u32 cnt_idx;          // index of class-A counter in the counters block (B, C follow)
};

struct Emit {             // where still-tied runs go (next round)
    u64* pool_rec;        // tiny pool: low 32 bits = suffix index
    u64* pool_hdr;        // sa_start | len << 32 | off << 40
    u64* seg_rec;         // segment array (record buffer `seg_buf`)
    u32 seg_buf;
    u32 pool_cnt_idx;     // counter indices
    u32 seg_cnt_idx;
    u32 pool_cap, seg_cap;
    u32 pool_chunk, seg_chunk;   // persistent workgroups reserve output room in chunks of this many slots
    u32 discard;                 // 1: the caller rebuilds next round's state itself (stateless doubling step): emit nothing
    u32 safe_rank;               // 1: LSD ranks from wave ballots (match-any); 0: from returning LDS atomics, which gfx950 serves
                                 // lowest lane first when lanes of one instruction meet on an address (measured:
                                 // tools/microbench/exp_lds_order.hip, 0 mismatches in 10^9) - not an architectural promise, so
                                 // every segment checks that it came out sorted and a violation repeats the round with ballots
    u32* grp_out;                // MODE_DEFER: tie-group head of every row (same indexing as the suffix-array rows)
    Lists lists;
};

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }

// XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// XCD and its L2); groups of T consecutive tiles feed the same output cursors, so give a whole group to one
// XCD: the adjacent runs its tiles claim then meet in ONE L2 and leave it as full lines.  Speed only.
__device__ __forceinline__ u32 xcd_tile(u32 b, u32 T)
{
    const u32 x = b & 7u, q = b >> 3;
    return ((q / T) * 8u + x) * T + (q % T);
}

// 16-byte records pairs.  Vector-memory instructions are what a CU runs out of first in the scatter kernels, so
// neighbouring records travel together wherever they can: one dwordx4 per lane instead of two dwordx2.
struct __attribute__((aligned(8))) Rec2 { u64 a, b; };
struct __attribute__((aligned(4))) Aux2 { u32 a, b; };      // the same for the records' 32-bit companions (RecBufs::x)

// inclusive wave scan with DPP row shifts / row broadcasts: 12 vector instructions, no LDS traffic (the __shfl_up form costs
// six ds_bpermute round trips - and the first wave of every scatter tile runs one while fifteen waves wait at a barrier)
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v)
{
    // within rows of 16 lanes: Kogge-Stone with row_shr 1, 2, 4, 8 (lanes without a source add 0)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    // across rows: lane 15 of the previous row to rows 1 and 3, then lane 31 to rows 2 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ u32 wave_excl_scan(u32 v, u32& total)
{
    const u32 x = wave_incl_scan_dpp(v);
    total = (u32)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}

// the same scan for max (unsigned, identity 0) and for bitwise or; lane 63 holds the wave's result
#define DPP_SCAN_STEP(OP, ctrl, rmask, bc) { const u32 o_ = (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, bc); v = OP; }
__device__ __forceinline__ u32 wave_incl_scan_max_dpp(u32 v)
{
    DPP_SCAN_STEP(o_ > v ? o_ : v, 0x111, 0xf, true) DPP_SCAN_STEP(o_ > v ? o_ : v, 0x112, 0xf, true)
    DPP_SCAN_STEP(o_ > v ? o_ : v, 0x114, 0xf, true) DPP_SCAN_STEP(o_ > v ? o_ : v, 0x118, 0xf, true)
    DPP_SCAN_STEP(o_ > v ? o_ : v, 0x142, 0xa, false) DPP_SCAN_STEP(o_ > v ? o_ : v, 0x143, 0xc, false)
    return v;
}
__device__ __forceinline__ u32 wave_or(u32 v)
{
    DPP_SCAN_STEP(o_ | v, 0x111, 0xf, true) DPP_SCAN_STEP(o_ | v, 0x112, 0xf, true)
    DPP_SCAN_STEP(o_ | v, 0x114, 0xf, true) DPP_SCAN_STEP(o_ | v, 0x118, 0xf, true)
    DPP_SCAN_STEP(o_ | v, 0x142, 0xa, false) DPP_SCAN_STEP(o_ | v, 0x143, 0xc, false)
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
#undef DPP_SCAN_STEP

__device__ __forceinline__ u32 wave_sum(u32 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// exclusive scan of 256 LDS values by the FIRST wave of the block (4 per lane); other waves skip.
// caller: __syncthreads() before (inputs ready) and after (outputs ready).
__device__ __forceinline__ u32 scan256_first_wave(const u32* in, u32* out)
{
    u32 total = 0;
    if (threadIdx.x < 64) {
        u32 l = threadIdx.x;
        u32 t0 = in[4 * l], t1 = in[4 * l + 1], t2 = in[4 * l + 2], t3 = in[4 * l + 3];
        u32 e = wave_excl_scan(t0 + t1 + t2 + t3, total);
        out[4 * l] = e; out[4 * l + 1] = e + t0; out[4 * l + 2] = e + t0 + t1; out[4 * l + 3] = e + t0 + t1 + t2;
    }
    return total;   // valid in wave 0 only
}

// the same for NB = 256 or 512 values (NB / 64 per lane)
template <int NB>
__device__ __forceinline__ u32 scanN_first_wave(const u32* in, u32* out)
{
    constexpr int PER = NB / 64;
    u32 total = 0;
    if (threadIdx.x < 64) {
        const u32 l = threadIdx.x;
        u32 v[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { v[i] = in[PER * l + i]; sum += v[i]; }
        u32 e = wave_excl_scan(sum, total);
#pragma unroll
        for (int i = 0; i < PER; ++i) { out[PER * l + i] = e; e += v[i]; }
    }
    return total;   // valid in wave 0 only
}

__device__ __forceinline__ u32 class_of(u32 len)   // 0:A 1:B 2:C 3:large   (len > TINY_MAX)
{
    return len <= CAP_A ? 0u : (len <= CAP_B ? 1u : (len <= CAP_C ? 2u : 3u));
}

__device__ __forceinline__ u64 pack_hdr(u32 sa_start, u32 len, u32 off)
{
    return (u64)sa_start | ((u64)len << 32) | ((u64)off << 40);
}

__device__ __forceinline__ void push_desc(const Lists& L, u32* counters, u32 cls, Desc d)
{
    u32 i = atomicAdd(&counters[L.cnt_idx + cls], 1u);
    if (i < L.cap[cls]) L.cls[cls][i] = d; else atomicOr(&counters[C_ERR], 1u);
}

// ------------------------------------------------------------------------------------------------
// 16-bit radix histogram (count_suffixes, cpp:1496-1521).  One workgroup per text chunk keeps ALL 65,536 bins in
// LDS as 16-bit counters, two per word (128 KiB): word = key' & 0x7fff, half-word = bit 15 of key', where
// key' = T[i] | T[i+1] << 8 is the key in MEMORY byte order (k_reduce16 transposes to the big-endian key the rest
// of the pipeline uses).  16 B per lane coalesced loads, next load in flight while the current one is counted,
// one non-returning LDS atomic per key.
//
// 16-bit counters can wrap.  Pass 0 counts the whole chunk without looking; every add contributes exactly 1 or
// 65536 to its word, so with true counts (L, H) the two stored halves sum to L + H - 65535 a - 65536 b with
// a = L / 65536 and b = (H + a) / 65536: the halves of all words add up to the number of keys of the chunk if and
// only if no counter wrapped.  When that check fails (a key with >= 65536 occurrences in one chunk: text, DNA,
// periodic data) pass 1 recounts the chunk in sub-chunks of H16_SUB keys: a counter below H16_FLUSH at the start of
// a sub-chunk cannot wrap inside it (H16_FLUSH + H16_SUB <= 65536), and after every sub-chunk a sweep over the LDS
// moves counters >= H16_FLUSH to an overflow list (at most chunk_len / H16_FLUSH entries ever), which is added to
// the output at the end.  partial[chunk][65536] is reduced by k_reduce16.
// ------------------------------------------------------------------------------------------------
#define H16_SUB 49152u       // bytes (= keys) per sub-chunk: 3 iterations of 1024 lanes x 16 B
#define H16_FLUSH 16384u
#define H16_OVF_CAP 1024u    // >= max chunk_len / H16_FLUSH (chunk_len <= 16 MiB)
#define H16_LDS_BYTES (131072u + H16_OVF_CAP * 8u + 64u)

// two bytes at byte offset o of the little-endian word stream w[]
__device__ __forceinline__ u32 h16_key(const u32* w, int o)
{
    const int s = o & 3;
    const u32 k = s == 3 ? __builtin_amdgcn_alignbyte(w[(o >> 2) + 1], w[o >> 2], 3) : (w[o >> 2] >> (8 * s));
    return k & 0xffffu;
}

__device__ __forceinline__ void h16_add(u32* h_lds, u32 k, u32 c)
{
    atomicAdd(&h_lds[k & 0x7fffu], c << ((k >> 15) * 16u));             // + c or + c << 16
}

// SUB = false: counts key' = bytes (j, j+1) of every position of the chunk.
// SUB = true:  counts key' = bytes (j+2, j+3) of the positions whose bytes (j, j+1) equal `sel` (memory order) - the
//              deeper histogram that lets a shard boundary fall INSIDE a heavy two-byte key ("if one key
//              exceeds n/G, refine that key with a deeper histogram"; the reference balances by handing out the largest
//              partitions first, cpp:1657-1678).
// Skewed chunks (pass 1, entered when a 16-bit counter wrapped in pass 0): a lane first merges runs of equal consecutive keys
// among its 16 positions into one add, and a wave whose lanes all add the same key and count issues ONE add for all of them -
// on a run of one repeated byte every LDS atomic of pass 0 is a 64-way same-address conflict (all-'A': 4.4 ms per GiB).
// MODE 2:      counts key' = bytes (j, j+1) of the positions whose bit is set in `bits` (one bit per text position): the
//              histograms of the B and B* suffixes for the two-stage build (induce_kernels.hip.h).
template <int MODE>
__global__ __launch_bounds__(1024) void k_hist16(const u8* __restrict__ text, u64 m, u32 chunk_len, u32 nchunks,
                                                 u32* __restrict__ partial, u32 sel, const unsigned short* __restrict__ bits,
                                                 u32 chunk0 /* first chunk of this launch: a rank of a multi-GPU job counts its stripes only */)
{
    constexpr bool SUB = MODE == 1, BITS = MODE == 2, FILT = MODE != 0;
    extern __shared__ u32 h_lds[];
    u64* ovf = reinterpret_cast<u64*>(h_lds + 32768);
    u32* ovf_n = h_lds + 32768 + H16_OVF_CAP * 2;            // [0] list length, [1] checksum, [2] keys counted (SUB), [3] largest counter
    // DENSE mode (round 5; small alphabets: DNA, plain text): a skewed chunk whose first sub-chunk showed at most 64 byte values does
    // not continue in SAFE mode (every add of a 4-letter text is a many-way same-address conflict, and SAFE mode's run merging and
    // sweeps cost 1.3 ms per GiB of DNA against 0.3 for random bytes) but counts DENSE keys code(b0) << S | code(b1) in up to 64
    // private copies of a small table (copy = lane: lanes never meet on an address), 32-bit counters, no sweeps; the counts of the
    // first sub-chunk are flushed to the output first, the dense totals are added to it at the end, and a byte value that only
    // turns up later is counted straight into the output (one global atomic, rare).
    __shared__ u32 s_present[8];
    __shared__ u8 s_dcode[256], s_dsym[256];
    const u32 chunk = blockIdx.x + chunk0, t = threadIdx.x;
    if (chunk >= nchunks) return;
    u32* const out = partial + (u64)blockIdx.x * 65536u;      // (a part's partials are indexed from its first chunk)
    u32 dense_s = 0, dense_lc = 0;                           // != 0: DENSE mode, S = bits per code; log2(copies)
    uint4* h4 = reinterpret_cast<uint4*>(h_lds);
    const u64 cbeg = (u64)chunk * chunk_len;
    u64 cend = cbeg + chunk_len;
    if (cend > m) cend = m;
    if (cend < cbeg) cend = cbeg;
    const u32 nsub = (u32)((cend - cbeg + H16_SUB - 1) / H16_SUB);
    // Optimistic counting (no sweeps, one plain add per key) is right for chunks whose keys are spread out.  Whether this is
    // such a chunk shows after the first sub-chunk, where no counter can have wrapped yet: if its largest counter, scaled to
    // the whole chunk, would pass 16 bits the chunk continues in SAFE mode right away instead of counting everything twice.
    bool safe = false;
#pragma unroll 1
    for (u32 pass = 0; pass < 2; ++pass) {
        for (u32 i = t; i < 8192u; i += 1024u) h4[i] = make_uint4(0, 0, 0, 0);
        if (t < 4) ovf_n[t] = 0;
        if (t < 8) s_present[t] = 0;
        __syncthreads();
        u64 base = cbeg + (u64)t * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        u32 nx = 0;
        u32 counted = 0;
        if (base < cend) { v = *reinterpret_cast<const uint4*>(text + base); nx = *reinterpret_cast<const u32*>(text + base + 16); }   // (the text is padded)
#pragma unroll 1
        for (u64 sub = cbeg; sub < cend; sub += H16_SUB) {
#pragma unroll 1
            for (u32 it = 0; it < H16_SUB / 16384u; ++it) {
                if (base >= cend) break;
                const uint4 cv = v;
                const u32 cn = nx;
                const u64 cb = base;
                base += 16384u;
                if (base < cend) { v = *reinterpret_cast<const uint4*>(text + base); nx = *reinterpret_cast<const u32*>(text + base + 16); }   // prefetch
                // (two loads in flight per lane instead of one: measured in round 6, 0.321 against 0.324 ms per GiB - the LDS atomics, not the
                // text, are what the kernel waits for: tools/microbench/exp_lds_hist_ceiling.hip)
                const u32 w[6] = {cv.x, cv.y, cv.z, cv.w, cn, 0u};
                const u32 lim = cend - cb >= 16 ? 16u : (u32)(cend - cb);
                u32 pb = 0xffffu;
                if (BITS) pb = bits[cb >> 4];
                if (dense_s) {
                    const u32 copy = lane_id() & ((1u << dense_lc) - 1u);
                    if constexpr (SUB) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            if ((u32)j < lim && h16_key(w, j) == sel) {
                                const u32 k = h16_key(w, j + 2);
                                const u32 c0 = s_dcode[k & 255u], c1 = s_dcode[k >> 8];
                                if ((c0 | c1) < 64u) atomicAdd(&h_lds[((((c0 << dense_s) | c1)) << dense_lc) | copy], 1u);
                                else atomicAdd(&out[k], 1u);                     // a byte value the first sub-chunk did not show
                            }
                        }
                    } else {
                        // (the second byte of position j is the first byte of position j + 1: one code look-up per byte)
                        u32 b0 = w[0] & 255u, c0 = s_dcode[b0];
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const u32 b1 = (w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 255u, c1 = s_dcode[b1];
                            bool on = (u32)j < lim;
                            if (BITS) on = on && ((pb >> j) & 1u);
                            if (on) {
                                if ((c0 | c1) < 64u) atomicAdd(&h_lds[((((c0 << dense_s) | c1)) << dense_lc) | copy], 1u);
                                else atomicAdd(&out[b0 | (b1 << 8)], 1u);        // a byte value the first sub-chunk did not show
                            }
                            b0 = b1; c0 = c1;
                        }
                    }
                } else
                if (!safe && !FILT && lim == 16u) {                // the common case: 16 plain adds, nothing predicated
#pragma unroll
                    for (int j = 0; j < 16; ++j) h16_add(h_lds, h16_key(w, j), 1u);
                } else if (!safe) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        bool on = (u32)j < lim;
                        if (SUB) on = on && h16_key(w, j) == sel;
                        if (BITS) on = on && ((pb >> j) & 1u);
                        if (on) { h16_add(h_lds, h16_key(w, SUB ? j + 2 : j), 1u); if (FILT) ++counted; }
                    }
                } else {
                    u32 run = 0, prev = 0;
#pragma unroll
                    for (int j = 0; j <= 16; ++j) {
                        bool on = (u32)j < lim && j < 16;
                        if (SUB && on) on = h16_key(w, j) == sel;
                        if (BITS && on) on = (pb >> j) & 1u;
                        if (FILT && on) ++counted;
                        const u32 k = on ? h16_key(w, SUB ? j + 2 : j) : 0u;
                        const bool flush = run != 0 && (!on || k != prev);
                        const u32 fk = prev, fc = run;
                        if (flush) run = 0;
                        if (on) { prev = k; ++run; }
                        const u64 fm = __ballot(flush);
                        if (fm) {
                            const u32 k0 = __builtin_amdgcn_readfirstlane(fk), c0 = __builtin_amdgcn_readfirstlane(fc);
                            if (__ballot(flush && (fk != k0 || fc != c0)) == 0) {          // the whole wave adds the same thing
                                if (flush && (int)lane_id() == __ffsll((long long)fm) - 1) h16_add(h_lds, k0, c0 * (u32)__popcll(fm));
                            } else if (flush) h16_add(h_lds, fk, fc);
                        }
                    }
                }
            }
            if (sub + H16_SUB >= cend) continue;                    // the last sub-chunk needs no sweep
            if (dense_s) continue;                                  // 32-bit private counters: nothing can wrap
            if (!safe && sub != cbeg) continue;                     // optimistic: only the look after the first sub-chunk
            __syncthreads();
            u32 cmax = 0;
#pragma unroll 1
            for (u32 i = t; i < 8192u; i += 1024u) {
                uint4 q = h4[i];
                if (!safe) {
                    const u32 a = max(max(q.x & 0xffffu, q.x >> 16), max(q.y & 0xffffu, q.y >> 16));
                    const u32 b = max(max(q.z & 0xffffu, q.z >> 16), max(q.w & 0xffffu, q.w >> 16));
                    cmax = max(cmax, max(a, b));
                    if (q.x | q.y | q.z | q.w) {                     // byte values seen so far (for the DENSE code): word = b0 | (b1 & 127) << 8, half = b1 >> 7
                        const u32 qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (!qq[e]) continue;
                            const u32 word = i * 4u + e, b0 = word & 255u, b1 = word >> 8;
                            atomicOr(&s_present[b0 >> 5], 1u << (b0 & 31u));
                            if (qq[e] & 0xffffu) atomicOr(&s_present[b1 >> 5], 1u << (b1 & 31u));
                            if (qq[e] >> 16) atomicOr(&s_present[(b1 | 128u) >> 5], 1u << (b1 & 31u));
                        }
                    }
                }
                if (((q.x | q.y | q.z | q.w) & 0xC000C000u) == 0) continue;
                u32 ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32 word = i * 4u + e;
                    if ((ww[e] & 0xffffu) >= H16_FLUSH) {
                        const u32 slot = atomicAdd(ovf_n, 1u);
                        if (slot < H16_OVF_CAP) { ovf[slot] = ((u64)word << 32) | (ww[e] & 0xffffu); ww[e] &= 0xffff0000u; }
                    }
                    if ((ww[e] >> 16) >= H16_FLUSH) {
                        const u32 slot = atomicAdd(ovf_n, 1u);
                        if (slot < H16_OVF_CAP) { ovf[slot] = ((u64)(word | 0x8000u) << 32) | (ww[e] >> 16); ww[e] &= 0xffffu; }
                    }
                }
                h4[i] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
            }
            if (!safe) {
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) cmax = max(cmax, (u32)__shfl_xor(cmax, sft, 64));
                if (lane_id() == 0) atomicMax(&ovf_n[3], cmax);
            }
            __syncthreads();
            // (counters >= H16_FLUSH were moved to the overflow list above in either mode, so switching is seamless)
            if (!safe && (u64)ovf_n[3] * nsub >= 65535ull) {
                safe = true;
                u32 sigma = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) sigma += (u32)__popc(s_present[k]);
                if (sigma >= 1u && sigma <= 64u) {                      // (workgroup-uniform) small alphabet: DENSE mode for the rest of the chunk
                    if (t < 256u) {
                        const u32 wd = s_present[t >> 5], bit = t & 31u;
                        u32 r = (u32)__popc(wd & ((1u << bit) - 1u));
                        for (u32 k = 0; k < (t >> 5); ++k) r += (u32)__popc(s_present[k]);
                        const bool on = (wd >> bit) & 1u;
                        s_dcode[t] = on ? (u8)r : (u8)255;
                        if (on) s_dsym[r] = (u8)t;
                    }
                    // what the first sub-chunk counted goes to the output now (the table is needed for the private copies)
                    for (u32 i = t; i < 32768u; i += 1024u) { const u32 q = h_lds[i]; out[i] = q & 0xffffu; out[i + 32768u] = q >> 16; h_lds[i] = 0; }
                    __threadfence();                                    // ... and is there before anything is ADDED to it
                    dense_s = sigma <= 2u ? 1u : 32u - (u32)__clz(sigma - 1u);
                    dense_lc = 15u - 2u * dense_s; if (dense_lc > 6u) dense_lc = 6u;
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (safe) break;
        // optimistic to the end: did any counter wrap after all?
        u32 sum = 0;
        for (u32 i = t; i < 8192u; i += 1024u) {
            const uint4 q = h4[i];
            sum += (q.x & 0xffffu) + (q.x >> 16) + (q.y & 0xffffu) + (q.y >> 16) + (q.z & 0xffffu) + (q.z >> 16) + (q.w & 0xffffu) + (q.w >> 16);
        }
        u32 listed = 0;                                             // what the look after the first sub-chunk moved to the list
        for (u32 i = t; i < min(ovf_n[0], H16_OVF_CAP); i += 1024u) listed += (u32)ovf[i];
        sum = wave_sum(sum + listed);
        if (FILT) counted = wave_sum(counted);
        if (lane_id() == 0) { atomicAdd(&ovf_n[1], sum); if (FILT) atomicAdd(&ovf_n[2], counted); }
        __syncthreads();
        const bool clean = ovf_n[1] == (FILT ? ovf_n[2] : (u32)(cend - cbeg));
        __syncthreads();
        if (clean) break;
        safe = true;                                                // recount everything with sweeps
    }
    if (dense_s) {
        // totals of the private copies (2^dense_lc consecutive words per dense key) -> added to the output under the key itself
        const u32 S = dense_s, LC = dense_lc, words = 1u << (2u * S + LC);
        for (u32 i = t; i < words; i += 1024u) {                   // (words is a multiple of 1024 or below it: whole waves stay together)
            u32 v = h_lds[i];
            for (u32 d = 1; d < (1u << LC); d <<= 1) v += (u32)__shfl_xor((int)v, (int)d, 64);
            if ((i & ((1u << LC) - 1u)) == 0 && v) {
                const u32 e = i >> LC;
                atomicAdd(&out[(u32)s_dsym[e >> S] | ((u32)s_dsym[e & ((1u << S) - 1u)] << 8)], v);
            }
        }
    } else
    for (u32 i = t; i < 32768u; i += 1024u) {
        const u32 q = h_lds[i];
        out[i] = q & 0xffffu;
        out[i + 32768u] = q >> 16;
    }
    const u32 no = ovf_n[0] < H16_OVF_CAP ? ovf_n[0] : H16_OVF_CAP;
    if (no) {                                           // uniform
        __threadfence();
        __syncthreads();
        for (u32 i = t; i < no; i += 1024u) atomicAdd(&out[(u32)(ovf[i] >> 32)], (u32)ovf[i]);
    }
}

// a part's partials -> 64-bit totals (the all-reduce's element type): 32-bit sums over blocks of 64 chunks (a chunk holds at most
// 2^24 keys), eight loads in flight per lane
__global__ __launch_bounds__(256) void k_reduce16_part(const u32* __restrict__ partial, u32 nchunks, u64* __restrict__ hist)
{
    const u32 kle = blockIdx.x * 256u + threadIdx.x;
    u64 total = 0;
    for (u32 c0 = 0; c0 < nchunks; c0 += 64u) {
        const u32 ce = min(c0 + 64u, nchunks);
        u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u32 c = c0;
        for (; c + 8 <= ce; c += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += partial[(u64)(c + j) * 65536u + kle];
        }
        for (; c < ce; ++c) s[0] += partial[(u64)c * 65536u + kle];
        total += (u64)s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
    }
    hist[((kle & 255u) << 8) | (kle >> 8)] = total;
}

// the all-reduced 64-bit histogram of a multi-GPU job (big-endian key order) -> the engine's own counters
template <bool W>
__global__ __launch_bounds__(256) void k_hist_from_u64(const u64* __restrict__ in, typename Wd<W>::hist_t* __restrict__ hist)
{
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    hist[k] = (typename Wd<W>::hist_t)in[k];
}

// sums the per-chunk partials (indexed by the memory-order key) and stores them under the big-endian key
template <bool W>
__global__ __launch_bounds__(256) void k_reduce16(const u32* __restrict__ partial, u32 nchunks, typename Wd<W>::hist_t* __restrict__ hist)
{
    // 256 workgroups (one per CU) x 256 keys; four independent partial sums keep four loads in flight per lane
    const u32 kle = blockIdx.x * 256u + threadIdx.x;           // T[i] | T[i+1] << 8
    typename Wd<W>::hist_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    u32 c = 0;
    for (; c + 4 <= nchunks; c += 4) {
        s0 += partial[(u64)c * 65536u + kle]; s1 += partial[(u64)(c + 1) * 65536u + kle];
        s2 += partial[(u64)(c + 2) * 65536u + kle]; s3 += partial[(u64)(c + 3) * 65536u + kle];
    }
    for (; c < nchunks; ++c) s0 += partial[(u64)c * 65536u + kle];
    hist[((kle & 255u) << 8) | (kle >> 8)] = s0 + s1 + s2 + s3;
}

// ------------------------------------------------------------------------------------------------
// 17-bit histogram for random-like inputs whose two-byte buckets outgrow the largest LDS sort (more than CAP_C records:
// uniform bytes from 1.15 GiB): key17 = the first 17 bits of the suffix.  With it the level-1 partition splits every
// first-byte segment 512 ways (k_partition<512>) and the children fit the bucket sort again - instead of a third pass
// over the records (k_count + k_partition: 24 bytes per suffix more, the 2x cliff of round 3).
// One workgroup per text chunk of H17_CHUNK bytes keeps all 131,072 bins in LDS as 8-BIT counters, four per word
// (128 KiB): the path is only taken for spread-out keys, where a bin sees ~H17_CHUNK / 2^17 = 8 keys per chunk.  A byte that
// wraps changes the sum of all bytes (by -256, or by -255 when it carries into its neighbour), so "sum of the bytes ==
// keys of the chunk" is an exact no-overflow test; a chunk that fails it raises `flag` and the build takes the
// three-level path.  partial[chunk] = the raw LDS words; k_reduce17 adds the bytes up.
// ------------------------------------------------------------------------------------------------
#define H17_CHUNK (1u << 20)       // upper bound of a chunk; the host cuts the text into a multiple of 256 chunks (one wave of workgroups
                                   // per CU and no ragged last round: 296 one-MiB chunks took two rounds, the second 16 % full)
#define H17_LDS_BYTES 131072u
// Chunks are the stripes of the level-0 scatter cut into q pieces (chunk = stripe * q + piece), so that the per-chunk first-byte
// sums fb[chunk][256] add up to what the scatter's stripe cursors need: when this histogram runs FIRST (sizes where the
// 17-bit levels are expected) it replaces the 16-bit pass altogether - its pair sums are the 16-bit histogram.
__global__ __launch_bounds__(1024) void k_hist17(const u8* __restrict__ text, u64 m, u32 stripe_len, u32 q, u32 sub_len, u32 nchunks, u32* __restrict__ partial,
                                                 u32* __restrict__ fb, u32* __restrict__ flag)
{
    extern __shared__ u32 h_lds[];
    const u32 chunk = blockIdx.x, t = threadIdx.x;
    if (chunk >= nchunks) return;
    uint4* h4 = reinterpret_cast<uint4*>(h_lds);
    for (u32 i = t; i < 8192u; i += 1024u) h4[i] = make_uint4(0, 0, 0, 0);
    __shared__ u32 s_sum;
    if (t == 0) s_sum = 0;
    __syncthreads();
    const u32 stripe = chunk / q, piece = chunk - stripe * q;
    u64 send = (u64)(stripe + 1) * stripe_len;                      // end of my stripe
    if (send > m) send = m;
    u64 cbeg = (u64)stripe * stripe_len + (u64)piece * sub_len;
    if (cbeg > send) cbeg = send;
    const u64 cend = cbeg + sub_len < send ? cbeg + sub_len : send;
    u64 base = cbeg + (u64)t * 16u;
    uint4 v = make_uint4(0, 0, 0, 0);
    u32 nx = 0;
    if (base < cend) { v = *reinterpret_cast<const uint4*>(text + base); nx = *reinterpret_cast<const u32*>(text + base + 16); }   // (the text is padded)
#pragma unroll 1
    for (; base < cend;) {
        const uint4 cv = v;
        const u32 cn = nx;
        const u64 cb = base;
        base += 16384u;
        if (base < cend) { v = *reinterpret_cast<const uint4*>(text + base); nx = *reinterpret_cast<const u32*>(text + base + 16); }   // prefetch
        const u32 w[5] = {cv.x, cv.y, cv.z, cv.w, cn};
        const u32 lim = cend - cb >= 16 ? 16u : (u32)(cend - cb);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 x = (j & 3) ? __builtin_amdgcn_alignbyte(w[(j >> 2) + 1], w[j >> 2], j & 3) : w[j >> 2];     // bytes j, j+1, j+2 (, j+3)
            const u32 k = ((x & 255u) << 9) | ((x >> 7) & 0x1feu) | ((x >> 23) & 1u);
            if ((u32)j < lim) atomicAdd(&h_lds[k >> 2], 1u << ((k & 3u) * 8u));
        }
    }
    __syncthreads();
    u32 sum = 0;
    uint4* out4 = reinterpret_cast<uint4*>(partial + (u64)chunk * 32768u);
    for (u32 i = t; i < 8192u; i += 1024u) {
        const uint4 q = h4[i];
        out4[i] = q;
        // bytes of four words summed pairwise: (q & 0x00ff00ff) + ((q >> 8) & 0x00ff00ff) holds two 16-bit sums per word
        const u32 a = (q.x & 0x00ff00ffu) + ((q.x >> 8) & 0x00ff00ffu), b = (q.y & 0x00ff00ffu) + ((q.y >> 8) & 0x00ff00ffu);
        const u32 c2 = (q.z & 0x00ff00ffu) + ((q.z >> 8) & 0x00ff00ffu), d = (q.w & 0x00ff00ffu) + ((q.w >> 8) & 0x00ff00ffu);
        const u32 e = a + b + c2 + d;
        const u32 quad = (e & 0xffffu) + (e >> 16);                 // keys counted in the 16 bins of this quad of words
        sum += quad;
        // first-byte sums: quad i covers bins 16 i .. 16 i + 15, first byte b = i / 32: the 32 lanes of a half wave hold the
        // 32 quads of ONE first byte (i = t + 1024 k: b = t / 32 + 32 k)
        u32 fbs = quad;
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) fbs += __shfl_xor(fbs, sft, 64);
        if ((t & 31u) == 0) fb[(u64)chunk * 256u + (i >> 5)] = fbs;
    }
    sum = wave_sum(sum);
    if (lane_id() == 0) atomicAdd(&s_sum, sum);
    __syncthreads();
    if (t == 0 && s_sum != (u32)(cend - cbeg)) atomicOr(flag, 1u);
}

// the 16-bit histogram from the 17-bit one (pairs of neighbouring bins)
__global__ __launch_bounds__(256) void k_pair16(const u32* __restrict__ hist17, u32* __restrict__ hist16)
{
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    const uint2 v = reinterpret_cast<const uint2*>(hist17)[k];
    hist16[k] = v.x + v.y;
}

// stripe sums of the level-0 scatter from the first-byte sums of the 17-bit histogram's chunks (unsharded builds: no key range)
__global__ __launch_bounds__(256) void k_stripe_sums17(const u32* __restrict__ fb, u32 q, u32* __restrict__ sums)
{
    const u32 stripe = blockIdx.x, b = threadIdx.x;
    u32 s = 0;
    for (u32 p = 0; p < q; ++p) s += fb[((u64)stripe * q + p) * 256u + b];
    sums[stripe * 256u + b] = s;
}

// hist17[k] = sum over the chunks of byte k of partial[chunk] (hist17 zeroed by the caller); blockIdx.y = group of chunks
__global__ __launch_bounds__(256) void k_reduce17(const u32* __restrict__ partial, u32 nchunks, u32 per_group, u32* __restrict__ hist17)
{
    const u32 word = blockIdx.x * 256u + threadIdx.x;        // 32768 words of four bins
    const u32 c0 = blockIdx.y * per_group;
    const u32 c1 = c0 + per_group < nchunks ? c0 + per_group : nchunks;
    u32 lo = 0, hi = 0;                                       // bins 0, 2 | bins 1, 3 as 16-bit sums (per_group <= 256 chunks x 255)
    u32 s[4] = {0, 0, 0, 0};
    for (u32 c = c0; c < c1; ++c) {
        const u32 q = partial[(u64)c * 32768u + word];
        lo += q & 0x00ff00ffu; hi += (q >> 8) & 0x00ff00ffu;
        if (((c - c0) & 255u) == 255u) { s[0] += lo & 0xffffu; s[2] += lo >> 16; s[1] += hi & 0xffffu; s[3] += hi >> 16; lo = 0; hi = 0; }
    }
    s[0] += lo & 0xffffu; s[2] += lo >> 16; s[1] += hi & 0xffffu; s[3] += hi >> 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (s[i]) atomicAdd(&hist17[word * 4u + i], s[i]);
}

// exclusive scan of the 131,072 counts -> start, cursor and count of every 17-bit bucket (record offsets of an unsharded
// build).  128 workgroups of 1024 bins: a workgroup first adds up everything in front of its bins (coalesced loads of counts
// that sit in the L2; the single-workgroup version with 128 consecutive bins per thread took 89 us), then scans its own.
// Cross-check against the 16-bit histogram the level-0 scatter was set up with (hist16 != nullptr): a mismatch raises `flag`.
__global__ __launch_bounds__(1024) void k_scan17(const u32* __restrict__ hist17, const u32* __restrict__ hist16,
                                                 u32* __restrict__ child_start, u32* __restrict__ child_cnt, u32* __restrict__ cursor1,
                                                 u32* __restrict__ flag, u32* __restrict__ hmax17)
{
    __shared__ u32 wsum[16];
    __shared__ u32 s_front;
    const u32 t = threadIdx.x, first = blockIdx.x * 1024u;
    if (t == 0) s_front = 0;
    __syncthreads();
    u32 front = 0;
    for (u32 k = t; k < first; k += 1024u) front += hist17[k];
    front = wave_sum(front);
    if (lane_id() == 0 && front) atomicAdd(&s_front, front);
    const u32 c = hist17[first + t];
    u32 wtot;
    const u32 e = wave_excl_scan(c, wtot);
    if (lane_id() == 63) wsum[t >> 6] = wtot;
    __syncthreads();
    u32 start = s_front + e;
    for (u32 k = 0; k < (t >> 6); ++k) start += wsum[k];
    child_start[first + t] = start;
    cursor1[first + t] = start;
    child_cnt[first + t] = c;
    if (hist16) {                                             // pairs of neighbouring bins are the 16-bit counts
        const u32 pair = c + (u32)__shfl_xor(c, 1, 64);
        if ((t & 1u) == 0 && pair != hist16[(first + t) >> 1]) atomicOr(flag, 2u);
    }
    u32 mx = c;
    for (int s2 = 32; s2 >= 1; s2 >>= 1) { const u32 o = __shfl_xor(mx, s2, 64); mx = o > mx ? o : mx; }
    if (lane_id() == 0 && mx) atomicMax(hmax17, mx);
}

// A shard works on its own key range: everything outside [klo, khi) reads as 0 (wide builds: the global counts are 64-bit,
// a shard's own counts and offsets fit 32 bits).  Boundary keys that the shard owns only in part are corrected by k_sub_fix.
template <bool W>
__global__ __launch_bounds__(256) void k_hist_clip(const typename Wd<W>::hist_t* __restrict__ hist, u32 klo, u32 khi, u32* __restrict__ hist32, u32* __restrict__ counters)
{
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    const u64 v = (k >= klo && k < khi) ? (u64)hist[k] : 0ull;
    if (v > 0xffffffffull) atomicOr(&counters[C_ERR], 0x100u);
    hist32[k] = (u32)v;
}

// Shard boundary inside the two-byte key `key` (big-endian): the shard owns only the suffixes whose NEXT two bytes lie in
// [sublo, subhi).  sub_partial[chunk][.] is the deeper histogram of that key (k_hist16<1>, memory byte order).  Per text
// chunk: the in-range count replaces hist_partial[chunk][key] (saved for k_sub_restore; the scatter's stripe cursors are made
// from these), and the total goes to hist_clip[key] (zeroed by the caller).
__global__ __launch_bounds__(256) void k_sub_fix(const u32* __restrict__ sub_partial, u32 sublo, u32 subhi, u32 key,
                                                 u32* __restrict__ hist_partial, u32* __restrict__ saved, u32* __restrict__ hist_clip_key)
{
    __shared__ u32 s_sum;
    const u32 chunk = blockIdx.x;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const u32* p = sub_partial + (u64)chunk * 65536u;
    u32 sum = 0;
    for (u32 sk = sublo + threadIdx.x; sk < subhi; sk += 256u) sum += p[(sk >> 8) | ((sk & 255u) << 8)];
    sum = wave_sum(sum);
    if (lane_id() == 0 && sum) atomicAdd(&s_sum, sum);
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 at = (u64)chunk * 65536u + ((key >> 8) | ((key & 255u) << 8));
        saved[chunk] = hist_partial[at];
        hist_partial[at] = s_sum;
        if (s_sum) atomicAdd(hist_clip_key, s_sum);
    }
}

__global__ __launch_bounds__(256) void k_sub_restore(u32* __restrict__ hist_partial, const u32* __restrict__ saved, u32 nchunks, u32 key)
{
    const u32 chunk = blockIdx.x * 256u + threadIdx.x;
    if (chunk < nchunks) hist_partial[(u64)chunk * 65536u + ((key >> 8) | ((key & 255u) << 8))] = saved[chunk];
}

#define SCAN16_LDS_BYTES ((32768u + 1024u + 32u) * 4u)
// Exclusive scan of the 65,536 counts (bucket offsets, cpp:1603-1630) + set-up of the two scatter
// levels for the key range [klo, khi) of this shard.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_scan16(const u32* __restrict__ hist, u32* __restrict__ bstart /*65537*/,
                                                 u32 klo, u32 khi,
                                                 u32* __restrict__ child_start, u32* __restrict__ child_cnt,
                                                 u32* __restrict__ cursor1, u32* __restrict__ seg0_base,
                                                 Desc* __restrict__ seg0, u32* __restrict__ tile_start0 /*257*/,
                                                 u32* __restrict__ counters, u32 z)
{
    __shared__ u32 wsum[16];
    __shared__ u32 s_in[256], s_out[256];
    __shared__ u32 s_half;
    extern __shared__ u32 tbuf[];                       // SCAN16_LDS_BYTES: 32768 words, one pad word per 32
    const u32 t = threadIdx.x;
    // A thread scans 32 CONSECUTIVE bins per half of the key space.  Reading them straight from global memory makes
    // every load instruction touch 64 cache lines; instead each half goes through LDS: coalesced global accesses on
    // one side, conflict-free rows (index i lives at i + i / 32) on the other.
#define SC_PAD(i) ((i) + ((i) >> 5))
    u32 v[2][32];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
#pragma unroll 8
        for (u32 k = 0; k < 32; ++k) tbuf[SC_PAD(k * 1024u + t)] = hist[h * 32768u + k * 1024u + t];
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < 32; ++k) v[h][k] = tbuf[t * 33u + k];
    }
    u32 ex[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32 loc = 0;
#pragma unroll
        for (u32 k = 0; k < 32; ++k) loc += v[h][k];
        u32 wtot;
        u32 e = wave_excl_scan(loc, wtot);
        __syncthreads();
        if (lane_id() == 63) wsum[t >> 6] = wtot;
        __syncthreads();
        if (t < 64) {
            u32 w = (t < 16) ? wsum[t] : 0, tot;
            u32 we = wave_excl_scan(w, tot);
            if (t < 16) wsum[t] = we;
            if (t == 0) { if (h == 0) s_half = tot; else bstart[65536] = s_half + tot; }
        }
        __syncthreads();
        ex[h] = wsum[t >> 6] + e + (h ? s_half : 0u);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        u32 run = ex[h];
#pragma unroll
        for (u32 k = 0; k < 32; ++k) { tbuf[t * 33u + k] = run; run += v[h][k]; }
        __syncthreads();
#pragma unroll 8
        for (u32 k = 0; k < 32; ++k) bstart[h * 32768u + k * 1024u + t] = tbuf[SC_PAD(k * 1024u + t)];
    }
#undef SC_PAD
    __threadfence_block();
    __syncthreads();
    const u32 base = bstart[klo];
    if (t < 256) {
        u32 lo = t << 8, hi = (t + 1) << 8;
        if (lo < klo) lo = klo;
        if (hi > khi) hi = khi;
        Desc d = {0, 0, 0, 0};
        if (lo < hi) { d.rec_off = bstart[lo] - base; d.len = bstart[hi] - bstart[lo]; d.sa_off = d.rec_off; }
        seg0[t] = d;
        seg0_base[t] = d.rec_off;
        s_in[t] = (d.len + P1_TILE - 1) / P1_TILE;
    }
    __syncthreads();
    u32 tt = scan256_first_wave(s_in, s_out);
    if (t == 0) {
        tile_start0[256] = tt;
        counters[C_MS] = bstart[khi] - base;
        counters[C_RANK0] = z + base;
    }
    __syncthreads();
    if (t < 256) tile_start0[t] = s_out[t];
}

// per-key outputs of the scan for the level-1 partition (child offsets, cursors, counts) + the two statistics the host
// reads (largest bucket, number of non-empty buckets): 64 workgroups instead of the tail of a single one
__global__ __launch_bounds__(1024) void k_scan16_post(const u32* __restrict__ hist, const u32* __restrict__ bstart, u32 klo, u32 khi,
                                                      u32* __restrict__ child_start, u32* __restrict__ child_cnt,
                                                      u32* __restrict__ cursor1, u32* __restrict__ counters)
{
    const u32 key = blockIdx.x * 1024u + threadIdx.x;
    const u32 base = bstart[klo];
    const bool in = key >= klo && key < khi;
    const u32 s = in ? bstart[key] - base : 0u;
    child_start[key] = s;
    cursor1[key] = s;
    const u32 c = in ? hist[key] : 0u;
    child_cnt[key] = c;
    u32 hmax = c;
    const u32 hnz = (u32)__popcll(__ballot(c != 0));
#pragma unroll
    for (int s2 = 32; s2 >= 1; s2 >>= 1) { const u32 o = __shfl_xor(hmax, s2, 64); hmax = o > hmax ? o : hmax; }
    __shared__ u32 s_nz, s_mx;                // (one pair of global atomics per workgroup: ~90 per us is all one address takes)
    if (threadIdx.x == 0) { s_nz = 0; s_mx = 0; }
    __syncthreads();
    if (lane_id() == 0) { if (hnz) atomicAdd(&s_nz, hnz); if (hmax) atomicMax(&s_mx, hmax); }
    __syncthreads();
    if (threadIdx.x == 0) { if (s_nz) atomicAdd(&counters[C_HNZ], s_nz); if (s_mx) atomicMax(&counters[C_HMAX], s_mx); }
}

// ------------------------------------------------------------------------------------------------
// Level-0 scatter: text -> records bucketed by FIRST byte (initial_two_byte_radix_sort, cpp:1525-1555,
// first half).  Record key = the 4 bytes that follow the first byte, so the next level needs no gather.
// Per tile: LDS histogram with returning LDS atomics gives the rank inside the tile, one global atomic
// per (tile, bin) claims the output range, records are staged bin-sorted in LDS and written as runs.
// ------------------------------------------------------------------------------------------------
// Level-0 cursors per text stripe: cursor0[stripe][b] = start of first-byte bucket b + number of in-range
// suffixes with first byte b in earlier stripes (from the per-chunk histograms k_hist16 left behind).  With one
// cursor set per stripe the 256 output streams become 256 x nchunks, which spreads the scatter's writes
// (and its claim atomics) over all HBM channels.
__global__ __launch_bounds__(256) void k_stripe_sums(const u32* __restrict__ partial, u32 per, u32 klo, u32 khi, u32* __restrict__ sums /* zeroed */)
{
    // sums[stripe][b] = in-range suffixes of this stripe (= `per` histogram chunks) whose first byte is b.
    // partial[] is indexed in memory byte order (first byte in the low 8 bits), so row b1 of a chunk is
    // contiguous over b: coalesced reads.  Four workgroups per stripe, one quarter of the second bytes each.
    const u32 c = blockIdx.x >> 2, q = blockIdx.x & 3u, b = threadIdx.x;
    u32 sum = 0;
    for (u32 h = 0; h < per; ++h) {
        const u32* p = partial + (u64)(c * per + h) * 65536u;
#pragma unroll 16
        for (u32 i = 0; i < 64u; ++i) {
            const u32 b1 = q * 64u + i;
            const u32 k = (b << 8) | b1;
            const u32 v = p[b1 * 256u + b];
            sum += (k >= klo && k < khi) ? v : 0u;
        }
    }
    if (sum) atomicAdd(&sums[c * 256u + b], sum);
}

// The same for SEVERAL key ranges at once (sharded histogram: what every shard of a multi-GPU job needs from the stripes this
// rank counted): the ranges are consecutive, keys.k[g] .. keys.k[g + 1] belongs to shard keys.first + g; one pass over the partials.
struct ShardKeys { u32 n, first; u32 k[65]; };
__global__ __launch_bounds__(256) void k_stripe_sums_multi(const u32* __restrict__ partial, u32 per, ShardKeys keys, u32* __restrict__ sums /* [shard][stripes_room][256], zeroed */,
                                                           u32 stripes_room)
{
    const u32 c = blockIdx.x >> 4, q = blockIdx.x & 15u, b = threadIdx.x;      // 16 workgroups per stripe, 16 second bytes each
    u32 g = 0, sum = 0;
    // thread b walks the keys b << 8 | b1 in rising order: its shard index only ever moves forward
    for (u32 i = 0; i < 16u; ++i) {
        const u32 b1 = q * 16u + i, k = (b << 8) | b1;
        if (k < keys.k[0] || k >= keys.k[keys.n]) continue;
        while (k >= keys.k[g + 1]) {
            if (sum) atomicAdd(&sums[((u64)(keys.first + g) * stripes_room + c) * 256u + b], sum);
            sum = 0; ++g;
        }
        for (u32 h = 0; h < per; ++h) sum += partial[(u64)(c * per + h) * 65536u + b1 * 256u + b];
    }
    if (sum) atomicAdd(&sums[((u64)(keys.first + g) * stripes_room + c) * 256u + b], sum);
}

__global__ __launch_bounds__(128) void k_stripes(const u32* __restrict__ sums, u32 nchunks,
                                                 const u32* __restrict__ seg0_base, u32* __restrict__ cursor0)
{
    __shared__ u32 wtot[2];
    const u32 b = blockIdx.x, c = threadIdx.x;
    const u32 sum = c < nchunks ? sums[c * 256u + b] : 0u;
    u32 wt;
    u32 e = wave_excl_scan(sum, wt);
    if (lane_id() == 63) wtot[c >> 6] = wt;
    __syncthreads();
    if (c >= 64) e += wtot[0];
    if (c < nchunks) cursor0[c * 256u + b] = seg0_base[b] + e;
}

// Prefix-doubling keys.  narrow: the rank itself (32 bits).  wide: ranks have up to 40 bits but a record only 24 key bits,
// so a doubling step sorts in two passes over the same (read-only) rank array - first on digit A = rank >> dig_shift,
// then, inside the groups that tie on it, on digit B = rank & dig_mask (the passes are ordinary rounds of the engine).
struct KeySpec {
    u64 depth;            // text: characters consumed so far; doubling: h
    u32 sigma, cpk, zlow; // text, dense code: alphabet size, symbols per key, left shift of the base-sigma number
    u32 dig_shift;        // doubling, wide: key24 = (rank >> dig_shift) & dig_mask
    u32 dig_mask;
};


// Text rounds: the sorts fetch the next key of a record THEMSELVES (records arrive with the suffix index only), instead of a
// separate k_refill pass in front of them.  The gathers are bound by random DRAM accesses (one sector per tied suffix:
// ~40 G/s chip-wide whatever the kernel), the sorts by LDS work: inside one kernel the waves that wait for their sectors
// leave the CU to the waves that sort (1 GiB text: k_refill 44 ms + LDS sorts 64 ms one after the other before).
// get_value of the reference (cpp:129-143): big-endian window at text + index + depth, zero beyond the end.
struct GatherSpec {
    const u8* text;       // nullptr: keys are already in the records
    u64 n;
    KeySpec ks;
    u32* pc_out;          // two-stage builds (narrow; nullptr: off): a sort that gathers a record's key ALSO reads the three characters in
                          // front of the suffix (pc_fetch: the same 128-byte line in four cases of five) and, when the record comes out
                          // FINAL (a tie run of one), leaves them at pc_out[row] next to the row - so that the induction's first level
                          // does not fetch them with a random text access per B* suffix (reference: the entries of the multi-threaded
                          // induction cache their preceding symbol, msufsort.cpp:674-790).  Written at most once per row, together with
                          // the row's final write; rows finalised by a kernel that gathered nothing keep PC_UNKNOWN.
    u32 flags;            // GS_TINY2: k_sort_tiny reads TWO keys' worth of symbols per gather and ranks by both (narrow text rounds whose
                          // two windows fit one 16-byte load): a run that the next round would have split is split now
};
#define GS_TINY2 1u
#define PC_UNKNOWN 0xffffffffu

// the (up to) three characters in front of suffix j and how many there are: T[j-1] | T[j-2] << 8 | T[j-3] << 16 | count << 24
__device__ __forceinline__ u32 pc_fetch(const u8* __restrict__ text, u32 j)
{
    if (j >= 4u) {
        u32 w;
        __builtin_memcpy(&w, text + j - 4u, 4);
        return (__builtin_bswap32(w) & 0xffffffu) | (3u << 24);
    }
    u32 v = 0;
    for (u32 k = 0; k < j && k < 3u; ++k) v |= (u32)text[j - 1u - k] << (8u * k);
    return v | ((j < 3u ? j : 3u) << 24);
}

// key of ONE suffix from its window; `code` = dense alphabet code (LDS), used when ks.cpk is not a plain window
template <bool W>
__device__ __forceinline__ u32 window_key(const u32* w /* 4 words, zero beyond what was loaded */, const KeySpec& ks, const u8* code)
{
    if (ks.cpk == (W ? 3u : 4u)) return __builtin_bswap32(w[0]);
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((u32)i < ks.cpk) acc = acc * ks.sigma + (u32)code[(w[i >> 2] >> (8 * (i & 3))) & 255u];
    return acc << ks.zlow;
}

// keys of up to N records of one lane, B gathers in flight at a time
template <bool W, int N, int B>
__device__ __forceinline__ void gather_keys(const GatherSpec& g, const u8* code, const typename Wd<W>::sa_t (&idx)[N], const bool (&valid)[N], u32 (&key)[N])
{
    const u32 wbytes = g.ks.cpk == (W ? 3u : 4u) ? 4u : (g.ks.cpk <= 8u ? 8u : 16u);      // kernel-uniform
#pragma unroll
    for (int b0 = 0; b0 < N; b0 += B) {
        u32 w[B][4];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            w[k][0] = w[k][1] = w[k][2] = w[k][3] = 0;
            if (b0 + k < N && valid[b0 + k]) {
                const u64 pos = (u64)idx[b0 + k] + g.ks.depth;
                if (pos < g.n) {                                   // (the text is padded with >= 64 zero bytes)
                    if (wbytes == 4u) __builtin_memcpy(w[k], g.text + pos, 4);
                    else if (wbytes == 8u) __builtin_memcpy(w[k], g.text + pos, 8);
                    else __builtin_memcpy(w[k], g.text + pos, 16);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < B; ++k) if (b0 + k < N) key[b0 + k] = valid[b0 + k] ? window_key<W>(w[k], g.ks, code) : 0xffffffffu;
    }
}

// k_sort_tiny with GS_TINY2 (narrow): this round's key AND the next round's from one 16-byte window (8 bytes for plain 4-byte windows).
// A tie run costs a random 64-byte sector per record and round whatever is read from it; the second key comes from the same sector
// in seven cases of eight.  Zero padding behind the text makes the second key what the next round would have gathered.
template <int N>
__device__ __forceinline__ void gather_keys2(const GatherSpec& g, const u8* code, const u32 (&idx)[N], const bool (&valid)[N], u32 (&key)[N], u32 (&key2)[N])
{
    const bool plain = g.ks.cpk == 4u;                          // (kernel-uniform; packed keys: 5 .. 8 symbols, GS_TINY2 is not set beyond)
    u32 w[N][4];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        w[k][0] = w[k][1] = w[k][2] = w[k][3] = 0;
        if (valid[k]) {
            const u64 pos = (u64)idx[k] + g.ks.depth;
            if (pos < g.n) {                                    // (the text is padded with >= 64 zero bytes)
                if (plain) __builtin_memcpy(w[k], g.text + pos, 8);
                else __builtin_memcpy(w[k], g.text + pos, 16);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (!valid[k]) { key[k] = 0xffffffffu; key2[k] = 0xffffffffu; continue; }
        key[k] = window_key<false>(w[k], g.ks, code);
        const u64 lo = (u64)w[k][0] | ((u64)w[k][1] << 32), hi = (u64)w[k][2] | ((u64)w[k][3] << 32);
        const u32 sh = 8u * g.ks.cpk;                           // 32 .. 64
        const u64 s2 = sh >= 64u ? hi : ((lo >> sh) | (hi << (64u - sh)));
        const u32 w2[4] = {(u32)s2, (u32)(s2 >> 32), 0u, 0u};
        key2[k] = window_key<false>(w2, g.ks, code);
    }
}

// number of dense key symbols of round 0 and where their base-sigma number sits in the key word (host and device agree):
// narrow: 3 symbols in the 24 bits below the bucket byte; wide: the 16 bits between the bucket byte and the index byte
// hold 3 symbols when sigma^3 <= 2^16, else 2
template <bool W> __host__ __device__ inline u32 s0_symbols(u32 sigma) { return (!W || (u64)sigma * sigma * sigma <= 65536ull) ? 3u : 2u; }
template <bool W> __host__ __device__ inline u32 s0_digit_bits(u32 sigma)      // bits of the largest dense number
{
    u64 p = 1;
    for (u32 k = 0; k < s0_symbols<W>(sigma); ++k) p *= sigma;
    u32 b = 0;
    while (b < 32 && ((p - 1) >> b) != 0) ++b;
    return b;
}

// Packing of the gather rounds' keys (host and device agree): cpk symbols read as ONE base-sigma number, shifted left by zlow;
// false: plain big-endian window of 4 (wide: 3) bytes.
template <bool W> __host__ __device__ inline bool key_packing(u32 sigma, u32 abits, u32& cpk, u32& zlow)
{
    cpk = W ? 3u : 4u; zlow = 0;
    if (abits < 2u || sigma < 2u) return false;
    u64 p = 1; u32 k = 0;
    while (k < 16 && p * sigma <= (1ull << (W ? 24 : 32))) { p *= sigma; ++k; }
    if (k < (W ? 4u : 5u)) return false;                   // not even one symbol more than a plain window
    u32 bl = 0;
    while (bl < 32 && ((p - 1) >> bl) != 0) ++bl;          // bits of the largest key
    cpk = k; zlow = 32 - bl;
    return true;
}

template <bool W>
__global__ __launch_bounds__(S0_THREADS) void k_scatter0(const u8* __restrict__ text, u64 m, u64 lo32, u64 hi32, u32 chunk_len,
                                                         u32* __restrict__ cursor0, u64* __restrict__ out,
                                                         const u8* __restrict__ code, const u32* __restrict__ counters, u32 allow_pack,
                                                         const u8* __restrict__ sel_bits /* one bit per position, or nullptr: all */,
                                                         u32* __restrict__ aux_out = nullptr /* narrow, small alphabets: key of the first gather round, per record */)
{
    // Small alphabets (up to 84 codes, k_alphabet): the three key symbols behind the second byte are written as ONE
    // dense base-sigma number, left-aligned in the 24 key bits below the bucket byte.  Same depth (5 characters), but
    // the partition levels below the two-byte buckets split on evenly used bits: one level instead of three for DNA
    // (sigma^3 = 125 values), two for a 28-letter text, and children that k_sort_fast2 accepts when the symbols are
    // evenly used.
    __shared__ u8 s_code[256];
    const u32 sigma = counters[C_ASIGMA];
    const bool tiny = allow_pack != 0u && sigma >= 2u && sigma <= 84u;        // kernel-uniform
    const u32 nsym = tiny ? s0_symbols<W>(sigma) : 3u;
    const u32 dshift = tiny ? 24u - s0_digit_bits<W>(sigma) : 0u;             // (wide: 8 + 16 - bits)
    // Compact staging: LDS holds the text of the tile and, per kept position, its 2-byte offset in the tile, bin-sorted; the
    // record (key bytes behind the position + index) is put together at write-out from the LDS copy of the text.  3 bytes of
    // LDS per position instead of 9, so a 16,384-position tile (runs of 64 records = 512 bytes per bin on uniform bytes) keeps two
    // workgroups per CU (52 KB each; runs beyond 512 bytes buy nothing at 256 bins: tools/microbench/exp_write_runs.hip).
    __shared__ __attribute__((aligned(16))) u32 tile[S0_TILE / 4 + 8];
    __shared__ __attribute__((aligned(16))) unsigned short spos[S0_TILE];
    __shared__ u32 hist[256], lstart[256], gbase[256];
    __shared__ u32 s_total;
    const u32 t = threadIdx.x;
    const u64 base0 = (u64)xcd_tile(blockIdx.x, chunk_len / S0_TILE) * S0_TILE;
    if (base0 >= m) return;
    const u64 base = base0 + (u64)t * S0_POS;
    if (t < 256) { hist[t] = 0; if (tiny) s_code[t] = code[t]; }
    __syncthreads();
    u32 w[S0_POS / 4 + 2];
#pragma unroll
    for (int k = 0; k < S0_POS / 4 + 2; ++k) w[k] = 0;
    if (base < m + S0_POS) {                     // (one thread past the end too: its words are the look-ahead of the last positions; the text is padded)
#pragma unroll
        for (int k = 0; k < S0_POS / 4 + 1; ++k) w[k] = *reinterpret_cast<const u32*>(text + base + 4 * k);
    }
#pragma unroll
    for (int k = 0; k < S0_POS / 4; ++k) tile[t * (S0_POS / 4) + k] = w[k];
    // look-ahead behind the tile: 4 bytes for the round-0 keys, up to 5 + 16 for the key of the first gather round (the text is
    // followed by MSUFSORT_HIP_TEXT_PAD = 64 zero bytes)
    if (t < 8) { const u64 at = base0 + S0_TILE + 4u * t; tile[S0_TILE / 4 + t] = at + 4 <= m + 64 ? *reinterpret_cast<const u32*>(text + at) : 0u; }
    u32 rank[S0_POS / 2];                       // two 16-bit ranks per word
    u32 validmask = 0;
    static_assert(S0_POS == 8 || S0_POS == 16, "the positions of one thread are one or two bytes of the bitmap");
    u32 selmask = S0_POS == 8 ? 0xffu : 0xffffu;
    if (sel_bits) selmask = base < m ? (S0_POS == 8 ? (u32)sel_bits[base >> 3] : (u32)reinterpret_cast<const u16*>(sel_bits)[base >> 4]) : 0u;
#pragma unroll
    for (int k = 0; k < S0_POS / 2; ++k) rank[k] = 0;
#pragma unroll
    for (int j = 0; j < S0_POS; ++j) {
        const u32 b0 = (w[j >> 2] >> (8 * (j & 3))) & 255u;
        const u32 b1 = (w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 255u;
        const u32 b2 = (w[(j + 2) >> 2] >> (8 * ((j + 2) & 3))) & 255u, b3 = (w[(j + 3) >> 2] >> (8 * ((j + 3) & 3))) & 255u;
        const u64 k32 = ((u64)b0 << 24) | (b1 << 16) | (b2 << 8) | b3;                 // shard = range of 4-byte prefixes
        const bool valid = (base + j < m) && k32 >= lo32 && k32 < hi32 && ((selmask >> j) & 1u);
        if (valid) { rank[j >> 1] |= atomicAdd(&hist[b0], 1u) << (16 * (j & 1)); validmask |= 1u << j; }
    }
    __syncthreads();
    // claim the output ranges now, consume the answer after staging (hides the atomic's latency)
    u32 claim = 0;
    if (t < 256) { const u32 c = hist[t]; if (c) claim = atomicAdd(&cursor0[(u32)(base0 / chunk_len) * 256u + t], c); }
    const u32 total = scan256_first_wave(hist, lstart);
    if (t == 0) s_total = total;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < S0_POS; ++j) {
        if (validmask & (1u << j)) {
            const u32 b0 = (w[j >> 2] >> (8 * (j & 3))) & 255u;
            spos[lstart[b0] + ((rank[j >> 1] >> (16 * (j & 1))) & 0xffffu)] = (unsigned short)(t * S0_POS + j);
        }
    }
    if (t < 256) gbase[t] = claim - lstart[t];            // slot -> output index is one add: out[gbase[bin] + slot]
    __syncthreads();
    const u32 tot = s_total;
    // key of the first gather round (window_key at depth 2 + nsym), read off the tile: that round then needs no random text access
    KeySpec ks1{};
    bool with_aux = false;
    if constexpr (!W) {
        if (aux_out != nullptr && tiny) {
            ks1.sigma = sigma; ks1.depth = 2u + nsym;
            with_aux = key_packing<W>(sigma, counters[C_ABITS], ks1.cpk, ks1.zlow);
        }
    }
    for (u32 s = t; s < tot; s += S0_THREADS) {
        const u32 l = spos[s];
        const u64 v = (((u64)tile[(l >> 2) + 1] << 32) | tile[l >> 2]) >> (8u * (l & 3u));      // text bytes l .. l + 4 (and more)
        const u32 b0 = (u32)v & 255u;
        u32 key;
        if (!tiny) key = __builtin_bswap32((u32)(v >> 8));                                       // (wide: the 4th byte is dropped by make_rec)
        else {
            u32 dg = 0;
#pragma unroll
            for (int q = 2; q <= 4; ++q)
                if ((u32)q < 2u + nsym) dg = dg * sigma + (u32)s_code[(u32)(v >> (8 * q)) & 255u];
            key = (((u32)(v >> 8) & 255u) << 24) | (dg << dshift);
        }
        const u32 o = gbase[b0] + s;
        out[o] = make_rec<W>(key, (typename Wd<W>::sa_t)(base0 + l));
        if constexpr (!W) {
            if (with_aux) {
                const u32 a = l + (u32)ks1.depth, sh = 8u * (a & 3u);
                const u32* tw = tile + (a >> 2);
                u32 w4[4] = {0, 0, 0, 0};
                const u32 t0 = tw[0], t1 = tw[1], t2 = tw[2];
                w4[0] = sh ? (t0 >> sh) | (t1 << (32u - sh)) : t0;
                w4[1] = sh ? (t1 >> sh) | (t2 << (32u - sh)) : t1;
                if (ks1.cpk > 8u) {
                    const u32 t3 = tw[3], t4 = tw[4];
                    w4[2] = sh ? (t2 >> sh) | (t3 << (32u - sh)) : t2;
                    w4[3] = sh ? (t3 >> sh) | (t4 << (32u - sh)) : t3;
                }
                aux_out[o] = window_key<W>(w4, ks1, s_code);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic multi-tile partition level for LARGE segments (the role multikey_quicksort's partition
// loop cpp:591-625 plays for big partitions): split every listed segment by one key byte.
// k_count: per-segment 256-bin histogram; k_segscan: child offsets + cursors; k_partition: scatter.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 find_seg(const u32* __restrict__ tile_start, u32 nseg, u32 tile)
{
    u32 lo = 0, hi = nseg;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (tile_start[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

// g.text != nullptr (first level of a text round): the records arrive without keys - gather them here and store them back
// for the levels and sorts that follow.
template <bool W>
__global__ __launch_bounds__(P1_THREADS) void k_count(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                      const u32* __restrict__ tile_start, u32 shift,
                                                      u32* __restrict__ seg_hist, GatherSpec g, const u8* __restrict__ code)
{
    __shared__ u32 hist[256];
    __shared__ u32 s_seg;
    __shared__ u8 s_code[256];
    const u32 t = threadIdx.x;
    const u32 tile = xcd_tile(blockIdx.x, 512u);
    if (tile >= tile_start[nseg]) return;
    if (t == 0) s_seg = find_seg(tile_start, nseg, tile);
    if (t < 256) { hist[t] = 0; if (g.text) s_code[t] = code[t]; }
    __syncthreads();
    const u32 s = s_seg;
    const Desc d = list[s];
    const u32 off = (tile - tile_start[s]) * P1_TILE;
    u64* src = bufs.p[d.buf & 3u] + d.rec_off;
    if (g.text) {
        typename Wd<W>::sa_t idx[P1_ITEMS];
        bool valid[P1_ITEMS];
        u32 key[P1_ITEMS];
#pragma unroll
        for (int j = 0; j < P1_ITEMS; ++j) { const u32 p = off + j * P1_THREADS + t; valid[j] = p < d.len; idx[j] = valid[j] ? rec_idx<W>(src[p]) : 0; }
        gather_keys<W, P1_ITEMS, 4>(g, s_code, idx, valid, key);
#pragma unroll
        for (int j = 0; j < P1_ITEMS; ++j) {
            const u32 p = off + j * P1_THREADS + t;
            if (valid[j]) { const u64 r = make_rec<W>(key[j], idx[j]); src[p] = r; atomicAdd(&hist[(u32)(r >> (32 + shift)) & 255u], 1u); }
        }
    } else {
#pragma unroll 4
    for (int j = 0; j < P1_ITEMS; ++j) {
        const u32 p = off + j * P1_THREADS + t;
        if (p < d.len) atomicAdd(&hist[(u32)(src[p] >> (32 + shift)) & 255u], 1u);
    }
    }
    __syncthreads();
    if (t < 256 && hist[t]) atomicAdd(&seg_hist[(u64)s * 256u + t], hist[t]);
}

__global__ __launch_bounds__(256) void k_segscan(const Desc* __restrict__ list, u32 nseg,
                                                 const u32* __restrict__ seg_hist,
                                                 u32* __restrict__ child_start, u32* __restrict__ cursor,
                                                 u32* __restrict__ trivial)
{
    __shared__ u32 s_in[256], s_out[256];
    const u32 s = blockIdx.x, t = threadIdx.x;
    if (s >= nseg) return;
    const Desc d = list[s];
    const u32 c = seg_hist[(u64)s * 256u + t];
    s_in[t] = c;
    __syncthreads();
    scan256_first_wave(s_in, s_out);
    __syncthreads();
    const u32 st = d.rec_off + s_out[t];
    child_start[(u64)s * 256u + t] = st;
    cursor[(u64)s * 256u + t] = st;
    if (c == d.len) trivial[s] = 1u;        // everything in one bin: the scatter is skipped
    if (t == 0 && d.len == 0) trivial[s] = 1u;
}

// AUX: every record has a companion word in bufs.x[] (the key of the first gather round, k_scatter0) that moves with it.  The
// tile's records leave first; the same LDS then stages (bin, companion) pairs at the same slots and the companions follow -
// no LDS beyond what the records need, so the two workgroups per CU stay.  `aux_sigma` points at counters[C_ASIGMA]: the
// level-1 launch is queued before the host knows whether the alphabet is small enough for k_scatter0 to have written companions.
template <int NB, bool AUX = false>
__global__ __launch_bounds__(P1_THREADS) void k_partition(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                          const u32* __restrict__ tile_start, u32 shift,
                                                          u32* __restrict__ cursor, const u32* __restrict__ trivial,
                                                          u32 alt0, u32 alt1, u32 alt2, const u32* __restrict__ aux_sigma = nullptr)
{
    __shared__ __attribute__((aligned(16))) u64 stage[P1_TILE];
    __shared__ u32 hist[NB], lstart[NB], gbase[NB];
    __shared__ u32 s_seg, s_total;
    const u32 t = threadIdx.x;
    const u32 tile = xcd_tile(blockIdx.x, 512u);
    if (tile >= tile_start[nseg]) return;
    if (t == 0) s_seg = find_seg(tile_start, nseg, tile);
    if (t < NB) hist[t] = 0;
    __syncthreads();
    const u32 s = s_seg;
    if (trivial && trivial[s]) return;
    const Desc d = list[s];
    const u32 off = (tile - tile_start[s]) * P1_TILE;
    const u64* src = bufs.p[d.buf & 3u] + d.rec_off;
    const u32 alt = (d.buf & 3u) == 0 ? alt0 : ((d.buf & 3u) == 1 ? alt1 : alt2);
    u64* dst = bufs.p[alt];
    bool aux_live = false;
    if constexpr (AUX) { const u32 sg = aux_sigma[0]; aux_live = sg >= 2u && sg <= 84u; }
    u64 rec[P1_ITEMS];
    u32 rank[P1_ITEMS];
    u32 ax[AUX ? P1_ITEMS : 1];
    static_assert(P1_ITEMS % 2 == 0, "records are read in pairs");
#pragma unroll
    for (int j = 0; j < P1_ITEMS; j += 2) {                  // items j, j+1 = records 2 (j/2 * THREADS + t) and the next one
        const u32 p = off + 2u * ((u32)(j / 2) * P1_THREADS + t);
        rec[j] = 0; rec[j + 1] = 0; rank[j] = 0xffffffffu; rank[j + 1] = 0xffffffffu;
        if constexpr (AUX) { ax[j] = 0; ax[j + 1] = 0; }
        if (p < d.len) {
            const Rec2 v = *reinterpret_cast<const Rec2*>(src + p);      // (may read one record past the segment)
            rec[j] = v.a; rec[j + 1] = v.b;
            if constexpr (AUX) {
                if (aux_live) { const Aux2 a2 = *reinterpret_cast<const Aux2*>(bufs.x[d.buf & 3u] + d.rec_off + p); ax[j] = a2.a; ax[j + 1] = a2.b; }
            }
            rank[j] = atomicAdd(&hist[(u32)(rec[j] >> (32 + shift)) & (u32)(NB - 1)], 1u);
            if (p + 1 < d.len) rank[j + 1] = atomicAdd(&hist[(u32)(rec[j + 1] >> (32 + shift)) & (u32)(NB - 1)], 1u);
        }
    }
    __syncthreads();
    u32 claim = 0;
    if (t < NB) { const u32 c = hist[t]; if (c) claim = atomicAdd(&cursor[(u64)s * (u32)NB + t], c); }
    const u32 total = scanN_first_wave<NB>(hist, lstart);
    if (t == 0) s_total = total;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < P1_ITEMS; ++j) {
        if (rank[j] != 0xffffffffu) {
            const u32 b = (u32)(rec[j] >> (32 + shift)) & (u32)(NB - 1);
            const u32 slot = lstart[b] + rank[j];
            stage[slot] = rec[j];
        }
    }
    if (t < NB) gbase[t] = claim - lstart[t];            // slot -> output index is one add: dst[gbase[bin] + slot]
    __syncthreads();
    const u32 tot = s_total;
    for (u32 q = t; q < tot; q += P1_THREADS) {
        const u64 r = stage[q];
        dst[gbase[(u32)(r >> (32 + shift)) & (u32)(NB - 1)] + q] = r;      // the bin is in the record itself
    }
    if constexpr (AUX) {
        if (!aux_live) return;                            // (kernel-uniform)
        u32* adst = bufs.x[alt];
        __syncthreads();                                  // the records have left the stage
#pragma unroll
        for (int j = 0; j < P1_ITEMS; ++j) {
            if (rank[j] != 0xffffffffu) {
                const u32 b = (u32)(rec[j] >> (32 + shift)) & (u32)(NB - 1);
                stage[lstart[b] + rank[j]] = ((u64)b << 32) | ax[j];
            }
        }
        __syncthreads();
        for (u32 q = t; q < tot; q += P1_THREADS) {
            const u64 r = stage[q];
            adst[gbase[(u32)(r >> 32)] + q] = (u32)r;
        }
    }
}

// Route the 256 children of every partitioned segment by size (partition scheduling, cpp:1652-1683,
// becomes a size-class dispatch): 1 -> final, 2..32 -> tiny pool, 33..CAP_C -> LDS-sort lists,
// larger -> next partition level.
template <bool W>
__global__ __launch_bounds__(256) void k_children(RecBufs bufs, const Desc* __restrict__ parents, u32 nseg,
                                                  const u32* __restrict__ child_start, const u32* __restrict__ child_cnt,
                                                  const u32* __restrict__ trivial, u32 alt0, u32 alt1, u32 alt2, u32 child_kbits,
                                                  typename Wd<W>::sa_t* __restrict__ sa_out, u32* __restrict__ isa, u32* __restrict__ grp_out, u32 mode,
                                                  u64* __restrict__ pool_rec, u64* __restrict__ pool_hdr, u32 pool_cnt_idx, u32 pool_cap,
                                                  Lists lists, Desc* __restrict__ lvl_dst, u32 lvl_cap, u32 lvl_cnt_idx, u32 lvl_tiles_idx,
                                                  u32* __restrict__ counters, u32 cbits = 8u /* log2(children per segment) */,
                                                  u32* __restrict__ pool_aux = nullptr /* companions of the pool records (bufs.x[]): see RecBufs */)
{
    const u64 c = (u64)blockIdx.x * 256u + threadIdx.x;
    const bool live = c < ((u64)nseg << cbits);
    const u32 s = live ? (u32)(c >> cbits) : 0u;
    const u32 cnt = live ? child_cnt[c] : 0u;
    Desc par = {0, 0, 0, 0};
    u32 start = 0, buf = 0, sa = 0;
    if (cnt) {
        par = parents[s];
        start = child_start[c];
        const bool triv = trivial && trivial[s];
        buf = triv ? (par.buf & 3u) : ((par.buf & 3u) == 0 ? alt0 : ((par.buf & 3u) == 1 ? alt1 : alt2));
        sa = par.sa_off + (start - par.rec_off);
    }
    const u64* src = bufs.p[buf] + start;
    bool aux_live = false;
    if constexpr (!W) { if (pool_aux != nullptr) { const u32 sg = counters[C_ASIGMA]; aux_live = sg >= 2u && sg <= 84u; } }
    const u32 rank0 = counters[C_RANK0];
    if (cnt == 1) {
        const auto idx = rec_idx<W>(src[0]);
        sa_out[sa] = idx;
        if (mode == MODE_ISA) isa[idx] = rank0 + sa + 1u;
        if (mode == MODE_DEFER) grp_out[sa] = sa;
    }
    {   // tiny children: one pool allocation per wave
        const u32 want = (cnt > 1 && cnt <= TINY_MAX) ? cnt : 0u;
        u32 wtot;
        const u32 woff = wave_excl_scan(want, wtot);
        u32 base = 0;
        if (wtot) {
            if (lane_id() == 0) base = atomicAdd(&counters[pool_cnt_idx], wtot);
            base = __shfl(base, 0, 64);
            if ((u64)base + wtot > pool_cap) { if (lane_id() == 0) atomicOr(&counters[C_ERR], 2u); }
            else if (want) {
                const u32 b = base + woff;
                const u64 st = (sa != par.sa_off || (par.buf & DESC_STALE)) ? (1ull << 48) : 0ull;      // header bit 48 = DESC_STALE of a pool run
                for (u32 k = 0; k < cnt; ++k) { pool_rec[b + k] = src[k]; pool_hdr[b + k] = pack_hdr(sa, cnt, k) | st; }
                if (aux_live) { const u32* asrc = bufs.x[buf] + start; for (u32 k = 0; k < cnt; ++k) pool_aux[b + k] = asrc[k]; }
            }
        }
    }
    // descriptors: one counter atomic per wave and class instead of one per child
    const u32 cls = cnt > TINY_MAX ? class_of(cnt) : 4u;
    const u64 lt_mask = lane_id() ? (~0ull >> (64 - lane_id())) : 0ull;
    const u32 stale = (cnt && (sa != par.sa_off || (par.buf & DESC_STALE))) ? DESC_STALE : 0u;
    const Desc d = {start, cnt, sa, DESC_BUF(child_kbits, buf | stale)};
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u64 m = __ballot(cls == k);
        if (m == 0) continue;
        const int leader = __ffsll((long long)m) - 1;
        u32 base = 0;
        if ((int)lane_id() == leader) base = atomicAdd(&counters[k < 3 ? lists.cnt_idx + k : lvl_cnt_idx], (u32)__popcll(m));
        base = __shfl(base, leader, 64);
        if (cls == k) {
            const u32 i = base + (u32)__popcll(m & lt_mask);
            if (k < 3) { if (i < lists.cap[k]) lists.cls[k][i] = d; else atomicOr(&counters[C_ERR], 1u); }
            else { if (i < lvl_cap) lvl_dst[i] = d; else atomicOr(&counters[C_ERR], 4u); }
        }
    }
    if (cls == 3) atomicAdd(&counters[lvl_tiles_idx], (cnt + P1_TILE - 1) / P1_TILE);
}

// exclusive scan of the tile counts of a large-segment list (single workgroup)
__global__ __launch_bounds__(1024) void k_tiles(const Desc* __restrict__ list, u32 nseg, u32* __restrict__ tile_start)
{
    __shared__ u32 wsum[16];
    __shared__ u32 s_carry;
    const u32 t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (u32 b = 0; b < nseg; b += 1024u) {
        const u32 i = b + t;
        const u32 v = i < nseg ? (list[i].len + P1_TILE - 1) / P1_TILE : 0u;
        u32 wt;
        const u32 e = wave_excl_scan(v, wt);
        if (lane_id() == 63) wsum[t >> 6] = wt;
        __syncthreads();
        u32 wbase = 0, tot = 0;
        for (u32 k = 0; k < 16; ++k) { if (k < (t >> 6)) wbase += wsum[k]; tot += wsum[k]; }
        const u32 carry = s_carry;
        if (i < nseg) tile_start[i] = carry + wbase + e;
        __syncthreads();
        if (t == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (t == 0) tile_start[nseg] = s_carry;
}

// Segments that stayed larger than CAP_C after all four key bytes: all keys equal, carry them to the next round
// unchanged.  k_carry_alloc reserves their room (one lane per segment), k_carry_copy moves the records by tiles
// (a 10^7-record segment must not be copied by one workgroup).
__global__ __launch_bounds__(256) void k_carry_alloc(const Desc* __restrict__ list, u32 nseg, u32* __restrict__ carry_base,
                                                     u32 seg_buf, u32 seg_cnt_idx, u32 seg_cap,
                                                     Desc* __restrict__ large_next, u32 large_cap, u32 large_cnt_idx, u32 large_tiles_idx,
                                                     u32* __restrict__ counters)
{
    const u32 s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    const Desc d = list[s];
    const u32 b = atomicAdd(&counters[seg_cnt_idx], d.len);
    carry_base[s] = 0xffffffffu;
    if ((u64)b + d.len > seg_cap) { atomicOr(&counters[C_ERR], 8u); return; }
    const u32 i = atomicAdd(&counters[large_cnt_idx], 1u);
    if (i >= large_cap) { atomicOr(&counters[C_ERR], 16u); return; }
    const Desc nd = {b, d.len, d.sa_off, seg_buf};
    large_next[i] = nd;
    atomicAdd(&counters[large_tiles_idx], (d.len + P1_TILE - 1) / P1_TILE);
    carry_base[s] = b;
}

template <bool W>
__global__ __launch_bounds__(P1_THREADS) void k_carry_copy(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                           const u32* __restrict__ tile_start, const u32* __restrict__ carry_base,
                                                           typename Wd<W>::sa_t* __restrict__ sa_out, u32* __restrict__ isa, u32* __restrict__ grp_out, u32 mode,
                                                           u64* __restrict__ seg_rec, const u32* __restrict__ counters, u32 with_aux = 0u)
{
    __shared__ u32 s_seg;
    const u32 t = threadIdx.x;
    if (blockIdx.x >= tile_start[nseg]) return;
    if (t == 0) s_seg = find_seg(tile_start, nseg, blockIdx.x);
    __syncthreads();
    const u32 s = s_seg;
    const u32 base = carry_base[s];
    if (base == 0xffffffffu) return;
    const Desc d = list[s];
    const u32 off = (blockIdx.x - tile_start[s]) * P1_TILE;
    const u64* src = bufs.p[d.buf & 3u] + d.rec_off;
    const u32 rank0 = counters[C_RANK0];
#pragma unroll
    for (int j = 0; j < P1_ITEMS; ++j) {
        const u32 p = off + j * P1_THREADS + t;
        if (p < d.len) {
            u64 r = src[p];
            if constexpr (!W) { if (with_aux) r = ((u64)bufs.x[d.buf & 3u][d.rec_off + p] << 32) | (u32)r; }      // keyed for the first gather round
            seg_rec[base + p] = r;
            sa_out[d.sa_off + p] = rec_idx<W>(r);
            if (mode == MODE_ISA) isa[(u32)r] = rank0 + d.sa_off + 1u;
            if (mode == MODE_DEFER) grp_out[d.sa_off + p] = d.sa_off;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Dense alphabet code.  Texts over small alphabets (DNA, plain ASCII) waste most of a 4-byte key window: with an
// order-preserving code of `bits` bits per symbol one 32-bit key holds 32 / bits symbols instead of 4, so every
// key-gather round resolves that many more characters per random text access.  code(0) = 0 (the zero padding past
// the end of the text must stay the smallest symbol), code(b) = 1 + number of smaller non-zero byte values that
// occur in the text.  The byte values that occur are the first bytes of the non-empty 16-bit buckets.
// ------------------------------------------------------------------------------------------------
template <bool W>
__global__ __launch_bounds__(256) void k_alphabet(const typename Wd<W>::hist_t* __restrict__ hist /* 65536, big-endian key, WHOLE text */, u8* __restrict__ code /* 256 */,
                                                  u32* __restrict__ counters)
{
    __shared__ u32 s_in[256], s_out[256];
    const u32 b = threadIdx.x;
    u32 any = 0;
    for (u32 k = 0; k < 256u; ++k) any |= hist[b * 256u + k] != 0 ? 1u : 0u;
    s_in[b] = (b != 0 && any != 0) ? 1u : 0u;
    __syncthreads();
    const u32 total = scan256_first_wave(s_in, s_out);
    __syncthreads();
    code[b] = b == 0 ? (u8)0 : (u8)(1u + s_out[b]);      // (codes of absent bytes are never looked up)
    if (b == 0) {
        const u32 ncodes = total + 1u;                   // + the reserved zero
        u32 bits = 1;
        while ((1u << bits) < ncodes) ++bits;
        counters[C_ABITS] = bits < 2u ? 2u : bits;
        counters[C_ASIGMA] = ncodes < 2u ? 2u : ncodes;
    }
}

// ------------------------------------------------------------------------------------------------
// Key refill for the next round (get_value, cpp:129-143): key = big-endian window of the text at depth `d`,
// zero beyond the end - 4 bytes, or (small alphabets) up to 16 symbols of the dense alphabet code read as one number
// in base sigma; in prefix-doubling rounds key = rank of suffix index + h (0 past n).
// ------------------------------------------------------------------------------------------------
template <bool W>
__device__ __forceinline__ u32 rank_key(const typename Wd<W>::sa_t* __restrict__ isa, u64 pos, u64 n, const KeySpec& ks)
{
    if (pos >= n) return 0u;                   // (the empty suffix has rank value 0 as well)
    if constexpr (W) return (u32)((isa[pos] >> ks.dig_shift) & ks.dig_mask) << 8;
    else return isa[pos];
}

template <bool W>
__global__ __launch_bounds__(256) void k_refill(u64* __restrict__ rec, const u32* __restrict__ counters, u32 cnt_idx,
                                                const u8* __restrict__ text, const typename Wd<W>::sa_t* __restrict__ isa,
                                                u64 n, u32 mode, const u8* __restrict__ code, KeySpec ks)
{
    // packed: key = (sum code(c_i) sigma^(cpk-1-i)) << zlow - the symbols read as ONE number in base sigma (denser than
    // bit fields: 13 DNA symbols per key instead of 10), left-aligned so that the top bits stay evenly used
    typedef typename Wd<W>::sa_t idx_t;
    __shared__ u8 s_code[256];
    const u32 sigma = ks.sigma, cpk = ks.cpk, zlow = ks.zlow;
    const u64 depth = ks.depth;
    const bool packed = mode == MODE_TEXT && cpk != (W ? 3u : 4u);
    if (packed) { s_code[threadIdx.x] = code[threadIdx.x]; __syncthreads(); }
    const u32 count = counters[cnt_idx];
    constexpr int U = 4;                       // independent gathers in flight per lane (latency bound otherwise)
    for (u64 base = (u64)blockIdx.x * 256u * U + threadIdx.x; base < count; base += (u64)gridDim.x * 256u * U) {
        idx_t idx[U];
        u32 key[U];
        bool v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const u64 i = base + (u64)k * 256u;
            v[k] = i < count; idx[k] = v[k] ? rec_idx<W>(rec[i]) : (idx_t)0;
            if constexpr (!W) v[k] = v[k] && idx[k] != 0xffffffffu;       // (records k_chain_resolve finished)
        }
        if (!packed) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const u64 pos = (u64)idx[k] + depth;
                key[k] = 0;
                if (v[k] && pos < n) {
                    if (mode == MODE_TEXT) {
                        u32 w;
                        __builtin_memcpy(&w, text + pos, 4);      // text is padded with >= 64 zero bytes
                        key[k] = __builtin_bswap32(w);             // (wide: make_rec keeps the first three bytes)
                    } else key[k] = rank_key<W>(isa, pos, n, ks);
                }
            }
        } else {
            u32 w[U][4];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const u64 pos = (u64)idx[k] + depth;
                w[k][0] = w[k][1] = w[k][2] = w[k][3] = 0;
                if (v[k] && pos < n) __builtin_memcpy(w[k], text + pos, 16);      // (pad: >= 64 zero bytes behind the text)
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                u32 acc = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((u32)i < cpk) acc = acc * sigma + (u32)s_code[(w[k][i >> 2] >> (8 * (i & 3))) & 255u];
                key[k] = acc << zlow;
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) if (v[k]) rec[base + (u64)k * 256u] = make_rec<W>(key[k], idx[k]);
    }
}

// Stateless doubling step (MODE_DEFER): the segment records of a shard are rebuilt from its suffix-array rows with
// identity placement - the record of (local) row r lives at rec[r] - and only for rows of groups larger than TINY_MAX
// (grp[head + TINY_MAX] == head); rows of smaller groups travel through the tiny pool (k_import_groups + k_refill).
// `act` (optional): the tied rows of the slice, in any order (ActiveSet in engine.hip); nullptr = all m rows.
template <bool W>
__global__ __launch_bounds__(256) void k_refill_rows(const typename Wd<W>::sa_t* __restrict__ sa_rows, const u32* __restrict__ grp, u32 m,
                                                     const u32* __restrict__ act, u32 nact,
                                                     u64* __restrict__ rec, const typename Wd<W>::sa_t* __restrict__ isa, u64 n, KeySpec ks)
{
    constexpr int U = 4;
    const u32 cnt = act ? nact : m;
    for (u64 base = (u64)blockIdx.x * 256u * U + threadIdx.x; base < cnt; base += (u64)gridDim.x * 256u * U) {
        typename Wd<W>::sa_t idx[U];
        u32 row[U];
        bool v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const u64 i = base + (u64)k * 256u;
            v[k] = false; idx[k] = 0; row[k] = 0;
            if (i < cnt) {
                const u32 r = act ? act[i] : (u32)i;
                row[k] = r;
                const u32 g = grp[r];
                v[k] = (u64)g + TINY_MAX < m && grp[g + TINY_MAX] == g;
                if (v[k]) idx[k] = sa_rows[r];
            }
        }
        u32 key[U];
#pragma unroll
        for (int k = 0; k < U; ++k) key[k] = v[k] ? rank_key<W>(isa, (u64)idx[k] + ks.depth, n, ks) : 0u;
#pragma unroll
        for (int k = 0; k < U; ++k) if (v[k]) rec[row[k]] = make_rec<W>(key[k], idx[k]);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS sort of one mid-size segment (33 .. THREADS*ITEMS records) - the GPU counterpart of
// multikey_quicksort + multikey_insertion_sort (cpp:488-642, 223-312) for one partition.
// LSD radix on the key bytes that actually differ inside the segment, 8 bits per pass, stable ranks from
// per-wave digit counters in LDS: a returning LDS atomic per record (lanes of one instruction that meet on an
// address are served lowest lane first on gfx950; every segment checks that it came out sorted) or, as the safe
// fallback, wave-wide digit matching with ballots (Emit::safe_rank).  In text rounds the records arrive with
// the suffix index only and the keys are gathered here (GatherSpec).  After sorting: rows are written to the
// suffix array, equal-key runs are detected with ballot bitmaps and the still-tied runs are compacted into next
// round's tiny pool / segment array.
// ------------------------------------------------------------------------------------------------
// Diagnostic build (-DMID_PROF): clock64() per phase of sort_mid_segment, summed by thread 0 of every workgroup into
// g_mid_prof[class][phase] (printed by the engine after every launch): 0 records loaded, 1 keys gathered, 2 sorted, 3 runs found +
// rows written, 4 room reserved, 5 emitted, 6 between segments; 8 segments, 9 records.
#ifdef MID_PROF
__device__ unsigned long long g_mid_prof[3][16];
#define MPROF(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); mprof[i] += now_ - mprof[15]; mprof[15] = now_; } } while (0)
#define MPROF_PARAM , unsigned long long* mprof
#define MPROF_ARG , mprof
#else
#define MPROF(i) do { } while (0)
#define MPROF_PARAM
#define MPROF_ARG
#endif

template <int THREADS, int ITEMS>
constexpr size_t sort_mid_lds_bytes()
{
    constexpr int CAP = THREADS * ITEMS;
    constexpr int WC = (THREADS / 64) * 256 + 8;               // per-wave digit counters
    return (size_t)CAP * 4 + (size_t)WC * 4 + 256 * 4 * 2 + (size_t)(CAP / 64) * (8 * 3 + 4 * 2) + 8 * 4 + 24 * 4;
}

// AUX (narrow, round 0 of small alphabets): every record has a companion word (RecBufs::x) - the key of the first gather round.
// It is permuted with the record (one more LDS exchange per LSD pass) and goes into the upper half of the records emitted for
// the next round, which therefore arrive WITH their key.
template <int THREADS, int ITEMS, bool W, bool AUX = false>
__device__ __forceinline__ void sort_mid_segment(const RecBufs& bufs, const Desc d,
                                                 typename Wd<W>::sa_t* __restrict__ sa_out, u32* __restrict__ isa, u32 mode,
                                                 const Emit& em, u32* __restrict__ counters, const GatherSpec& g, const u8* s_code MPROF_PARAM)
{
    constexpr u32 KL = klow<W>();                   // low bits of the key word that are not key (wide: index bits 32..39)
    // Output room (tiny pool, segment array, descriptor lists) is reserved from the global counters in CHUNKS
    // kept in LDS (ach[]): one returning global atomic per chunk instead of per segment - on text-like input
    // millions of segments emit tie runs every round and the shared counters were the bottleneck.
    // Unused chunk tails are overwritten with neutral entries (len 0) that every consumer skips.
    constexpr int CAP = THREADS * ITEMS;
    constexpr int NWV = THREADS / 64;
    constexpr int NW = CAP / 64;                    // bitmap words
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WCNT = NWV * 256 + 8;                           // LSD per-wave digit counters
    u32* ex = reinterpret_cast<u32*>(smem_raw);                 // CAP
    u32* wcnt = ex + CAP;                                       // WCNT
    u32* tot = wcnt + WCNT;                                     // 256
    u32* dbase = tot + 256;                                     // 256
    u64* bm_eq = reinterpret_cast<u64*>(dbase + 256);           // NW
    u64* bm_tiny = bm_eq + NW;                                  // NW
    u64* bm_seg = bm_tiny + NW;                                 // NW
    u32* pre_tiny = reinterpret_cast<u32*>(bm_seg + NW);        // NW
    u32* pre_seg = pre_tiny + NW;                               // NW
    u32* misc = pre_seg + NW;                                   // 8
    u32* ach = misc + 8;                                        // 10 chunk words + 14 pending fills (see k_sort_mid)

    const u32 len = d.len;
    if (len == 0) return;                           // neutral list entry (unused chunk tail)
    MPROF(6);
#ifdef MID_PROF
    if (threadIdx.x == 0) { mprof[8] += 1; mprof[9] += len; }
#endif
    // (the thread id is re-materialised per segment: the address arithmetic hanging off it is then redone here instead
    // of being hoisted out of the caller's segment loop into registers that cost a whole wave of occupancy)
    u32 t;
    asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(threadIdx.x));
    const u32 lane = t & 63u, wv = t >> 6;
    // every wave takes the same number of consecutive 64-record rows (wave-major order keeps the LSD passes stable);
    // with a fixed ITEMS rows per wave a short segment would be sorted by one wave while the others idle
    const u32 rpw = ((len + 63u) / 64u + NWV - 1) / NWV;            // rows per wave, <= ITEMS because len <= CAP
    const u32 wbase = wv * 64u * rpw;
    const u64* src = bufs.p[d.buf & 3u] + d.rec_off;
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;

    u32 key[ITEMS], idx[ITEMS], pos[ITEMS];
    u32 ax[AUX ? ITEMS : 1];
    u32 diff = 0;
    if (g.text == nullptr) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 p = wbase + j * 64 + lane;
            key[j] = 0xffffffffu; idx[j] = 0xffffffffu;
            if constexpr (AUX) ax[j] = 0;
            if ((u32)j < rpw && p < len) {
                const u64 r = src[p]; key[j] = (u32)(r >> 32); idx[j] = (u32)r;
                if constexpr (AUX) ax[j] = bufs.x[d.buf & 3u][d.rec_off + p];
            }
        }
    } else {        // text round: the records carry the suffix index only, the key is gathered here
        typename Wd<W>::sa_t fi[ITEMS];
        bool valid[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 p = wbase + j * 64 + lane;
            valid[j] = (u32)j < rpw && p < len;
            fi[j] = valid[j] ? rec_idx<W>(src[p]) : 0;
        }
#ifdef MID_PROF
        asm volatile("" :: "v"(fi[0]));
        MPROF(0);
#endif
        if constexpr (AUX) {        // (gather rounds of a two-stage build: the companion slot carries the characters in front of the suffix)
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) ax[j] = (valid[j] && g.pc_out) ? pc_fetch(g.text, (u32)fi[j]) : PC_UNKNOWN;
        }
        gather_keys<W, ITEMS, (ITEMS > 8 ? 6 : 8)>(g, s_code, fi, valid, key);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            idx[j] = valid[j] ? (u32)fi[j] : 0xffffffffu;
            if constexpr (W) { if (valid[j]) key[j] = (key[j] & 0xffffff00u) | (u32)(fi[j] >> 32); }      // the index byte rides in the key word
        }
    }
    // block-wide OR of the key differences (against the segment's first key, which thread 0 holds)
    if (t == 0) tot[0] = key[0];
    if (t < 8) misc[t] = 0;
    __syncthreads();
    {
        const u32 key0 = tot[0];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { const u32 p = wbase + j * 64 + lane; if ((u32)j < rpw && p < len) diff |= key[j] ^ key0; }
    }
    diff = wave_or(diff);
    MPROF(1);
    if (lane == 0 && diff) atomicOr(&misc[0], diff);
    __syncthreads();
    diff = misc[0] & (0xffffffffu << KL);           // (wide: the index byte never decides an LSD pass; equal keys keep their order)
    // number of item rows this wave takes part in (rows with at least one valid element)
    int rows = 0;
    if (wbase < len) { const u32 rem = len - wbase; rows = (int)((rem + 63u) / 64u); if (rows > (int)rpw) rows = (int)rpw; }

    const u32 rank0 = counters[C_RANK0];
    u32 rs[ITEMS], rl[ITEMS];        // run start / run length of each of my (sorted) elements
    bool any_eq = false;
    // One-wave segments of up to 128 records (the bulk of class A on text): all-pairs counting instead of LSD passes.
    const bool small = (THREADS == 64) && len <= 128u;
    if (!small) {
    const bool sorted_done = (diff == 0);
#pragma nounroll
    for (u32 shift = 0; shift < 32 && !sorted_done; shift += 8) {
        if (((diff >> shift) & 255u) == 0) continue;            // byte equal everywhere: pass not needed
        for (u32 i = t; i < (u32)NWV * 256u; i += THREADS) wcnt[i] = 0;
        __syncthreads();
        if (!em.safe_rank) {
            // stable rank inside the wave = what a returning LDS atomic hands back: instructions of a wave are served in
            // order, same-address lanes of one instruction lowest lane first
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) pos[j] = atomicAdd(&wcnt[wv * 256u + ((key[j] >> shift) & 255u)], 1u);
        } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (j < rows) {
                const u32 dg = (key[j] >> shift) & 255u;
                u64 mask = ~0ull;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const bool bit = (dg >> b) & 1u;
                    const u64 bal = __ballot(bit);
                    mask &= bit ? bal : ~bal;
                }
                const int leader = __ffsll((long long)mask) - 1;
                u32 old = 0;
                if ((int)lane == leader) old = atomicAdd(&wcnt[wv * 256u + dg], (u32)__popcll(mask));
                old = __shfl(old, leader, 64);
                pos[j] = old + (u32)__popcll(mask & lt_mask);
            }
        }
        }
        __syncthreads();
        for (u32 dg = t; dg < 256u; dg += THREADS) {
            u32 run = 0;
#pragma unroll
            for (int w = 0; w < NWV; ++w) { const u32 c = wcnt[w * 256 + dg]; wcnt[w * 256 + dg] = run; run += c; }
            tot[dg] = run;
        }
        __syncthreads();
        scan256_first_wave(tot, dbase);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (j < rows) {
                const u32 dg = (key[j] >> shift) & 255u;
                pos[j] += dbase[dg] + wcnt[wv * 256u + dg];
                ex[pos[j]] = key[j];
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) if (j < rows) key[j] = ex[wbase + j * 64 + lane];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) if (j < rows) ex[pos[j]] = idx[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) if (j < rows) idx[j] = ex[wbase + j * 64 + lane];
        __syncthreads();
        if constexpr (AUX) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) if (j < rows) ex[pos[j]] = ax[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) if (j < rows) ax[j] = ex[wbase + j * 64 + lane];
            __syncthreads();
        }
    }

    MPROF(2);
    // ---- equal-key runs ----
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) if (j < rows) ex[wbase + j * 64 + lane] = key[j];
    for (u32 i = t; i < (u32)NW; i += THREADS) { bm_eq[i] = 0; bm_tiny[i] = 0; bm_seg[i] = 0; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (j < rows) {
            const u32 p = wbase + j * 64 + lane;
            const bool eqn = (p + 1 < len) && ((key[j] >> KL) == (ex[p + 1] >> KL));
            if ((p + 1 < len) && ((key[j] >> KL) > (ex[p + 1] >> KL))) atomicOr(&counters[C_ERR], 0x200u);      // not sorted: the LDS-atomic ranks
                                                                                                               // were not stable after all
            const u64 bal = __ballot(eqn);
            if (lane == 0) bm_eq[p >> 6] = bal;
            any_eq |= (bal != 0);
        }
    __syncthreads();
    // Run boundaries without per-element walks: the first wave scans the bitmap words once -
    // pre_tiny[w] = 1 + last position with a 0 bit in words < w (start of a run that enters word w from the left),
    // pre_seg[w]  = first position with a 0 bit in words > w (end of a run that leaves word w to the right).
    // (pre_tiny/pre_seg are reused for the compaction prefix sums further down.)
    if (t < 64) {
        constexpr int PER = (NW + 63) / 64;
        u32 vs[PER], ve[PER];
        u32 mx = 0, mn = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int w = (int)t * PER + k;
            const u64 z = w < NW ? ~bm_eq[w] : ~0ull;
            vs[k] = z ? (u32)((w << 6) + 64 - __clzll((long long)z)) : 0u;
            ve[k] = z ? (u32)((w << 6) + __ffsll((long long)z) - 1) : 0xffffffffu;
            mx = vs[k] > mx ? vs[k] : mx;
            mn = ve[k] < mn ? ve[k] : mn;
        }
        // inclusive scans over lanes: prefix max, suffix min (= complement of the prefix max of the complements, lanes reversed):
        // DPP scans, three lane permutations instead of fourteen
        const u32 pmx = wave_incl_scan_max_dpp(mx);
        const u32 smn = ~(u32)__shfl(wave_incl_scan_max_dpp(~(u32)__shfl(mn, 63 - (int)lane, 64)), 63 - (int)lane, 64);
        u32 run_s = __shfl_up(pmx, 1, 64); if (lane == 0) run_s = 0;                 // exclusive prefix max
        u32 run_e = __shfl_down(smn, 1, 64); if (lane == 63) run_e = 0xffffffffu;    // exclusive suffix min
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int w = (int)t * PER + k;
            if (w < NW) pre_tiny[w] = run_s;
            run_s = vs[k] > run_s ? vs[k] : run_s;
        }
#pragma unroll
        for (int k = PER - 1; k >= 0; --k) {
            const int w = (int)t * PER + k;
            if (w < NW) pre_seg[w] = run_e;
            run_e = ve[k] < run_e ? ve[k] : run_e;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (j < rows) {
            const u32 p = wbase + j * 64 + lane;
            rs[j] = p; rl[j] = 1;
            if (p < len) {
                const u32 w = p >> 6, bit = p & 63u;
                const u64 z = ~bm_eq[w];
                const u64 below = bit ? (z & (~0ull >> (64 - bit))) : 0ull;      // 0 bits strictly below me
                const u64 above = z & (~0ull << bit);                              // 0 bits at or above me
                const u32 s = below ? (u32)((w << 6) + 64 - __clzll((long long)below)) : pre_tiny[w];
                const u32 e = above ? (u32)((w << 6) + __ffsll((long long)above) - 1) : pre_seg[w];
                rs[j] = s; rl[j] = e - s + 1;
                sa_out[d.sa_off + p] = full_idx<W>(key[j], idx[j]);
                if constexpr (AUX) { if (g.text && g.pc_out && rl[j] == 1u) g.pc_out[d.sa_off + p] = ax[j]; }
                if (mode == MODE_ISA && (s != 0 || (d.buf & DESC_STALE))) isa[idx[j]] = rank0 + d.sa_off + s + 1u;
                if (mode == MODE_DEFER) em.grp_out[d.sa_off + p] = d.sa_off + s;
            }
        }
    } else {
        // rank = #smaller keys + #equal keys before me; the equal keys are my tie run.  Every lane reads the same
        // LDS word per step (broadcast), no barriers inside the loop.
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) { const u32 p = j * 64 + lane; if (p < len) ex[p] = key[j]; }
        __syncthreads();
        u32 c_lt[2] = {0, 0}, c_eq[2] = {0, 0}, c_eqb[2] = {0, 0};
#define SMALL_CMP(kword, qq) do { const u32 kk = (kword) >> KL; _Pragma("unroll") for (int j = 0; j < 2; ++j) { const bool e = kk == (key[j] >> KL); \
            c_lt[j] += kk < (key[j] >> KL); c_eq[j] += e; c_eqb[j] += e & ((qq) < (u32)j * 64u + lane); } } while (0)
        const u32 len4 = len & ~3u;
        for (u32 q = 0; q < len4; q += 4) {              // four keys per LDS round trip (every lane reads the same 16 bytes)
            const uint4 k4 = *reinterpret_cast<const uint4*>(ex + q);
            SMALL_CMP(k4.x, q); SMALL_CMP(k4.y, q + 1u); SMALL_CMP(k4.z, q + 2u); SMALL_CMP(k4.w, q + 3u);
        }
        for (u32 q = len4; q < len; ++q) SMALL_CMP(ex[q], q);
#undef SMALL_CMP
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u32 p = j * 64 + lane;
            if (p < len) { const u32 f = c_lt[j] + c_eqb[j]; ex[f] = idx[j]; ex[128 + f] = c_lt[j]; ex[256 + f] = c_eq[j]; if (W) ex[384 + f] = key[j]; if constexpr (AUX) ex[384 + f] = ax[j]; }
        }
        for (u32 i = t; i < (u32)NW; i += THREADS) { bm_eq[i] = 0; bm_tiny[i] = 0; bm_seg[i] = 0; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 p = j * 64 + lane;
            rs[j] = p; rl[j] = 1;
            if (j < 2 && p < len) {
                idx[j] = ex[p]; rs[j] = ex[128 + p]; rl[j] = ex[256 + p];
                if (W) key[j] = ex[384 + p];             // (the index byte travels with the key word)
                if constexpr (AUX) ax[j] = ex[384 + p];
                sa_out[d.sa_off + p] = full_idx<W>(key[j], idx[j]);
                if constexpr (AUX) { if (g.text && g.pc_out && rl[j] == 1u) g.pc_out[d.sa_off + p] = ax[j]; }
                if (mode == MODE_ISA && (rs[j] != 0 || (d.buf & DESC_STALE))) isa[idx[j]] = rank0 + d.sa_off + rs[j] + 1u;
                if (mode == MODE_DEFER) em.grp_out[d.sa_off + p] = d.sa_off + rs[j];
                any_eq |= rl[j] > 1;
            }
        }
    }
    MPROF(3);
    if (em.discard) return;                         // stateless doubling step: the caller rebuilds the groups from grp_out
    if (!__syncthreads_or(any_eq)) return;

    // ---- compact still-tied runs into next round's structures ----
    u32 hc[3] = {0, 0, 0};           // run heads of this wave per size class
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (j < rows) {
            const u32 p = wbase + j * 64 + lane;
            const bool tied = (p < len) && rl[j] > 1;
            const u64 bt = __ballot(tied && rl[j] <= TINY_MAX);
            const u64 bs = __ballot(tied && rl[j] > TINY_MAX);
            if (lane == 0) { bm_tiny[p >> 6] = bt; bm_seg[p >> 6] = bs; }
            if (bs) {        // run heads per size class = descriptors this segment will push (exact, so none are wasted)
                const bool head = tied && rl[j] > TINY_MAX && p == rs[j];
                const u32 cls = class_of(rl[j]);
#pragma unroll
                for (u32 k = 0; k < 3; ++k) hc[k] += (u32)__popcll(__ballot(head && cls == k));
            }
        }
    if (lane == 0) {
#pragma unroll
        for (u32 k = 0; k < 3; ++k) if (hc[k]) atomicAdd(&misc[4 + k], hc[k]);
    }
    __syncthreads();
    if (t < 64) {
        u32 ct = 0, cs = 0;
        constexpr int PER = (NW + 63) / 64;
        u32 lt_[PER], ls_[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int w = (int)t * PER + k;
            lt_[k] = w < NW ? (u32)__popcll(bm_tiny[w]) : 0u;
            ls_[k] = w < NW ? (u32)__popcll(bm_seg[w]) : 0u;
            ct += lt_[k]; cs += ls_[k];
        }
        u32 tt, ts;
        u32 et = wave_excl_scan(ct, tt);
        u32 es = wave_excl_scan(cs, ts);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int w = (int)t * PER + k;
            if (w < NW) { pre_tiny[w] = et; pre_seg[w] = es; }
            et += lt_[k]; es += ls_[k];
        }
        if (t == 0) {
            u32 bt = 0, bs = 0, bad = 0;
            u32* pf = ach + 10;                       // pending fills: (begin, end) x {pool, seg, desc A, desc B, desc C}
#pragma unroll
            for (int k = 0; k < 10; ++k) pf[k] = 0;
            if (tt) {
                if (ach[0] + tt > ach[1]) {
                    pf[0] = ach[0]; pf[1] = ach[1];
                    const u32 sz = tt > em.pool_chunk ? tt : em.pool_chunk;
                    const u32 b = atomicAdd(&counters[em.pool_cnt_idx], sz);
                    // (a chunk past the end of the buffer is never adopted: the workgroup lives on and would
                    // hand its "room" to later segments)
                    if ((u64)b + sz > em.pool_cap) { atomicOr(&counters[C_ERR], 32u); bad = 1; ach[0] = 0; ach[1] = 0; }
                    else { ach[0] = b; ach[1] = b + sz; }
                }
                if (!bad) { bt = ach[0]; ach[0] += tt; }
            }
            if (ts) {
                if (ach[2] + ts > ach[3]) {
                    pf[2] = ach[2]; pf[3] = ach[3];
                    const u32 sz = ts > em.seg_chunk ? ts : em.seg_chunk;
                    const u32 b = atomicAdd(&counters[em.seg_cnt_idx], sz);
                    if ((u64)b + sz > em.seg_cap) { atomicOr(&counters[C_ERR], 64u); bad = 1; ach[2] = 0; ach[3] = 0; }
                    else { ach[2] = b; ach[3] = b + sz; }
                }
                if (!bad) { bs = ach[2]; ach[2] += ts; }
                // descriptor room: exactly the run heads counted above, taken from a chunk per class; a request that
                // does not fit gives up less than it asks for, so the lists never need more than twice the descriptors
                // that can exist
#pragma unroll
                for (u32 k = 0; k < 3; ++k) {
                    const u32 need = misc[4 + k];
                    if (need && ach[4 + 2 * k] + need > ach[5 + 2 * k]) {
                        pf[4 + 2 * k] = ach[4 + 2 * k]; pf[5 + 2 * k] = ach[5 + 2 * k];
                        // chunk per DESTINATION class: a neutral descriptor left over in a chunk costs its consumer a
                        // whole (empty) iteration, which is cheap for the 64-thread class-A sort and expensive for the
                        // 1024-thread class-C sort; exact mode (chunk 0): no slack at all
                        const u32 dch = em.seg_chunk ? (k == 0 ? 64u : (k == 1 ? 16u : 4u)) : 0u;
                        const u32 sz = need > dch ? need : dch;
                        const u32 b = atomicAdd(&counters[em.lists.cnt_idx + k], sz);
                        if ((u64)b + sz > em.lists.cap[k]) { atomicOr(&counters[C_ERR], 1u); bad = 1; ach[4 + 2 * k] = 0; ach[5 + 2 * k] = 0; }
                        else { ach[4 + 2 * k] = b; ach[5 + 2 * k] = b + sz; }
                    }
                }
            }
            misc[1] = bt; misc[2] = bs; misc[3] = bad;
        }
    }
    __syncthreads();
    MPROF(4);
    if (misc[3]) return;
    // neutral-fill the chunk tails that were just abandoned
    {
        const u32* pf = ach + 10;
        for (u32 i = pf[0] + t; i < pf[1]; i += THREADS) { em.pool_rec[i] = 0; em.pool_hdr[i] = 0; }
        for (u32 i = pf[2] + t; i < pf[3]; i += THREADS) em.seg_rec[i] = 0;
#pragma unroll
        for (u32 k = 0; k < 3; ++k)
            for (u32 i = pf[4 + 2 * k] + t; i < pf[5 + 2 * k]; i += THREADS) { const Desc z = {0, 0, 0, 0}; em.lists.cls[k][i] = z; }
    }
    const u32 base_t = misc[1], base_s = misc[2];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (j < rows) {
            const u32 p = wbase + j * 64 + lane;
            if (p < len && rl[j] > 1) {
                const u32 w = p >> 6;
                if (rl[j] <= TINY_MAX) {
                    const u32 o = base_t + pre_tiny[w] + (u32)__popcll(bm_tiny[w] & lt_mask);
                    u64 r = (u64)full_idx<W>(key[j], idx[j]);
                    if constexpr (AUX) { if (!g.text) r |= (u64)ax[j] << 32; }        // next round's key comes with the record
                    em.pool_rec[o] = r;
                    em.pool_hdr[o] = pack_hdr(d.sa_off + rs[j], rl[j], p - rs[j]);
                } else {
                    const u32 o = base_s + pre_seg[w] + (u32)__popcll(bm_seg[w] & lt_mask);
                    u64 r = (u64)full_idx<W>(key[j], idx[j]);
                    if constexpr (AUX) { if (!g.text) r |= (u64)ax[j] << 32; }
                    em.seg_rec[o] = r;
                    if (p == rs[j]) {
                        const Desc nd = {o, rl[j], d.sa_off + rs[j], em.seg_buf};
                        const u32 cls = class_of(rl[j]);
                        em.lists.cls[cls][atomicAdd(&ach[4 + 2 * cls], 1u)] = nd;      // room was reserved above
                    }
                }
            }
        }
    MPROF(5);
}

#include "bucket_sort_bits.hip.h"

// ------------------------------------------------------------------------------------------------
// Fast LDS sort for segments whose keys are spread out (the two-byte buckets of round 0 on random-like
// input) - the hot kernel of rounds 1 and 2; since round 3 k_sort_bits (bucket_sort_bits.hip.h) is tried first and this
// one stays selectable (MSUFSORT_HIP_BUCKET_SORT=fast2).  Persistent workgroups stride over the list.
//   1. one MSD split on the top BITS of the kbits varying key bits into 2^BITS sub-buckets; the rank r
//      inside the sub-bucket comes from a returning LDS atomic (arrival order, arbitrary);
//   2. block-wide exclusive scan of the sub-bucket counts;
//   3. every record stores the composite c = (varying key bits << 6) | r at base + r.  Composites are
//      distinct and globally ordered like (key, r), so a record's FINAL row is base + #{c' < c} over the
//      next FAST_PROBE slots from its sub-bucket's base - straight-line code, no loop, no predicate: slots
//      of later sub-buckets hold larger composites and the tail is padded with 0xffffffff.  The number of
//      slots with (c' ^ c) < 64 is the length of the record's tie run (equal keys).
//      Sub-buckets longer than FAST_PROBE (rare) finish in a short wave-uniform loop;
//   4. suffix indices are exchanged through LDS and written to the suffix array coalesced; the few tie
//      runs are compacted for the next round.
// Needs kbits <= 26 (composite in 32 bits) and no sub-bucket above FAST_LIMIT (r < 64); other segments are
// handed back to k_sort_mid through fb_list.
//
// Register budget = memory/compute overlap.  A CU running 1024 threads gives each 128 VGPRs.  Key, index, base,
// row and run information of 18 records per thread would fill them, leave nothing to keep the NEXT segment's
// records in flight, and make every CU alternate between a memory phase and a compute phase of about the same
// length.  So
//   - suffix indices wait in LDS (exi[source position], written and read by the same thread), not in registers,
//   - the sub-bucket table shares its LDS with the composite array (it is dead once every row holds its base),
//   - base, size and arrival rank share one register per record; the composite is read back from LDS by its owner,
//   - run information exists only for the few records that need it: they are pushed to a small tie list in LDS,
// which leaves room for the next segment's records: their loads are issued a few at a time behind the barriers of
// the sort (a CU accepts vector-memory instructions slowly; eighteen back to back stall every wave) and arrive
// while this segment is sorted.  Records travel as 16-byte pairs, rows leave as aligned 8-byte stores.
// Segments with more than TL tied records go back to k_sort_mid (after their rows have been written once).
// Diagnostic build: -DFAST2_PROF accumulates clock64() per phase (printed by the engine after the launch).
// ------------------------------------------------------------------------------------------------
#define FAST_LIMIT 48u
#ifndef FAST_PROBE
#define FAST_PROBE 5
#endif
#ifndef FAST_BITS_C
#define FAST_BITS_C 15      // sub-buckets of the class-C instance: 2 per record of a full segment
#endif
#ifndef FAST_BITS_B
#define FAST_BITS_B 13
#endif
#ifdef FAST2_PROF
__device__ unsigned long long g_fast2_prof[16];
#define F2P(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); prof_acc[i] += now_ - prof_last; prof_last = now_; } } while (0)
#else
#define F2P(i) do { } while (0)
#endif
#define FAST2_TL_C 768       // tie-list capacity of the class-C instance
#define FAST2_TL_B 256       // ... of the class-B instance (four workgroups per CU: 40,320 B of LDS each)
#ifndef FAST2_C_ITEMS
#define FAST2_C_ITEMS CLS_C_ITEMS    // records per thread of the class-C instance (even: they are loaded in pairs)
#endif
template <int THREADS, int ITEMS, int BITS, int TL>
__global__ __launch_bounds__(THREADS) void k_sort_fast2(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                        u32* __restrict__ sa_out, u32* __restrict__ isa, u32 mode,
                                                        Emit em, u32* __restrict__ counters, u32* __restrict__ fb_list, u32 fb_cnt_idx)
{
    constexpr int CAP = THREADS * ITEMS;
    constexpr int W = THREADS / 64;
    constexpr int NBIN = 1 << BITS;
    // Sub-bucket table: one 8-byte entry per FOUR sub-buckets = {their counts, one byte each | base of the first}.
    // Counts never pass FAST_LIMIT in a segment that is kept, so a byte is enough; the scan then reads an eighth of
    // the words a count-per-word table would need.
    constexpr int NW = NBIN / 4;
    constexpr int EW = NW / THREADS;
    constexpr u32 TRASH_POS = CAP + 32;
    constexpr u32 NWP = NW;
    constexpr u32 TRASH_W = NWP;
    static_assert(2 * NWP + 16 <= CAP + 64, "the sub-bucket table must fit under the composite array");
    static_assert((EW & (EW - 1)) == 0 && EW >= 1 && (THREADS & (THREADS - 1)) == 0, "power-of-two shapes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32* ex = reinterpret_cast<u32*>(smem_raw);                 // CAP + 64: sub-bucket table, composites, index exchange
    uint2* tab = reinterpret_cast<uint2*>(ex);
    u32* exi = ex + CAP + 64;                                   // CAP: suffix index of source position p
    u32* tl = exi + CAP;                                        // 3 * (u32)TL: tie list {index, rs | rl << 16 | ro << 24, local offset}
    u32* tot = tl + 3 * (u32)TL;                               // 16
    u32* misc = tot + 16;                                       // 16

    // Every lane owns PAIRS of neighbouring records: items 2q, 2q+1 of thread t are positions 2 (q THREADS + t) and
    // the next one, so records arrive as 16-byte loads and indices move through LDS as 8-byte accesses (half the
    // vector-memory instructions of an 8-byte-per-lane layout; their issue rate is what the CU runs out of first).
    static_assert(ITEMS % 2 == 0, "items come in pairs");
    constexpr int NL = ITEMS / 2;
    u32 t = threadIdx.x;
#define FAST_P(j) (((((u32)(j) >> 1) * (u32)THREADS + t) << 1) + ((u32)(j) & 1u))
    // table entry w lives at tab[FAST_W(w)]: thread t scans the EW consecutive entries t*EW .. t*EW+EW-1, which this map
    // puts at k*THREADS + t - consecutive lanes touch consecutive entries (no bank conflicts in the scan)
#define FAST_W(w) ((((w) & (u32)(EW - 1)) * (u32)THREADS) | ((w) / (u32)EW))
#define FAST_SCAN_AT(k) ((u32)(k) * (u32)THREADS + t)
#define BYTESUM(x) __builtin_amdgcn_sad_u8((x), 0u, 0u)        // sum of the four bytes: one full-rate instruction (a 32-bit multiply is quarter rate)
    u32 seg = blockIdx.x;
    if (seg >= nseg) return;
    // descriptors are fetched two segments ahead and the record buffer is picked with selects, not with an indexed
    // load from the argument block: no scalar-memory round trip sits in front of the record loads
#define FAST_SRC(dd) (reinterpret_cast<const unsigned char*>((((dd).buf & 3u) == 0u ? bufs.p[0] : ((dd).buf & 3u) == 1u ? bufs.p[1] : bufs.p[2]) + (dd).rec_off))
    Desc d = list[seg];
    Desc dn = list[seg + gridDim.x < nseg ? seg + gridDim.x : seg];      // unconditional, clamped: stays a scalar load
    u64 nrec[ITEMS];
    {
        // uniform base + 32-bit byte offset per lane (one address register per load, not two); no branches: the
        // offset is clamped instead
        const unsigned char* src = FAST_SRC(d);
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const u32 p = FAST_P(2 * q);
            const Rec2 v = *reinterpret_cast<const Rec2*>(src + (p < d.len ? p * 8u : 0u));    // (may read one record past the segment)
            nrec[2 * q] = v.a; nrec[2 * q + 1] = v.b;
        }
    }
    const u32 rank0 = counters[C_RANK0];
#ifdef FAST2_PROF
    __shared__ unsigned long long prof_acc[16];
    unsigned long long prof_last = clock64();
    if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) prof_acc[i] = 0;
#endif
    for (;;) {
        // the thread id is re-materialised per segment so that the address arithmetic hanging off it is redone
        // here (a handful of adds) instead of being hoisted out of the loop into registers this kernel does not have
        asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(threadIdx.x));
        const u32 lane = t & 63u, wv = t >> 6;
        const u32 len = d.len, sa_off = d.sa_off, cur = seg, kbits = (d.buf >> 8) & 255u;
        const bool stale = (d.buf & DESC_STALE) != 0;
        u32 key[ITEMS];                    // key, then composite, then base | size << 16 | rank << 24, then final row
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { key[j] = (u32)(nrec[j] >> 32); exi[FAST_P(j)] = (u32)nrec[j]; }
        F2P(10);
        seg += gridDim.x;
        const bool more = seg < nseg;
        d = dn;
        // The next segment's records travel while this one is sorted.  A CU can only keep so many requests in
        // flight: 18 loads per wave issued back to back stall every wave in the issue queue for microseconds, so
        // they are issued three rows at a time, one batch behind each phase of the sort (F2_LOAD).
        const unsigned char* nsrc = FAST_SRC(d);
        const u32 nlen = more ? d.len : 0u;
#define F2_LOAD(from, upto) do { _Pragma("unroll") for (int q_ = (from); q_ < (upto) && q_ < NL; ++q_) { const u32 p_ = FAST_P(2 * q_); const Rec2 v_ = *reinterpret_cast<const Rec2*>(nsrc + (p_ < nlen ? p_ * 8u : 0u)); nrec[2 * q_] = v_.a; nrec[2 * q_ + 1] = v_.b; } } while (0)
        constexpr int LB = (NL + 5) / 6 > 1 ? (NL + 5) / 6 : 2;  // pair loads per batch, up to six batches
        F2_LOAD(0, LB);
        dn = list[seg + gridDim.x < nseg ? seg + gridDim.x : nseg - 1u];
        F2P(0);
        const u32 nrows = (len + 127u) >> 7;                     // a wave covers 128 consecutive positions per pair load
        const int rows = 2 * (nrows > wv ? (int)((nrows - wv + W - 1) / W) : 0);     // items of this wave that can be inside the segment
        const u32 kmask = kbits >= 32 ? 0xffffffffu : ((1u << kbits) - 1u);
        const u32 sh = kbits > (u32)BITS ? kbits - BITS : 0u;
        bool ok = kbits <= 26u && len != 0 && len <= (u32)CAP;   // (longer class members go to k_sort_mid)
        u32 res_t = 0, res_s = 0;

        if (ok) {                                                // ---- phase 1: clear the table
            if (t < 16) misc[t] = 0;
            uint4* h4 = reinterpret_cast<uint4*>(ex);
            const uint4 z4 = {0u, 0u, 0u, 0u};
            for (u32 i = t; i < (2u * NWP + 16u) / 4u; i += THREADS) h4[i] = z4;
            F2P(11);
            __syncthreads();                                                        // (1)
            F2P(1);
        }
        F2_LOAD(LB, 2 * LB);
        if (ok) {                                                // ---- phase 2: arrival ranks
            bool skew = false;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) {
                    const u32 p = FAST_P(j);
                    const u32 k = key[j] & kmask;
                    const u32 g = k >> sh, by = (g & 3u) * 8u;
                    const u32 r = (atomicAdd(&tab[p < len ? FAST_W(g >> 2) : TRASH_W].x, 1u << by) >> by) & 255u;
                    skew |= (p < len) & (r >= FAST_LIMIT);                  // (a byte cannot wrap before somebody sees this)
                    key[j] = p < len ? ((k << 6) | (r & 63u)) : 0xffffffffu;
                }
            if (skew) misc[5] = 1u;
            __syncthreads();                                                        // (2)
            if (misc[5]) ok = false;
            F2P(2);
        }
        F2_LOAD(2 * LB, 3 * LB);
        if (ok) {                                                // ---- phase 3: scan -> base of every table entry
            u32 sum = 0;
#pragma unroll
            for (int k = 0; k < EW; ++k) sum += BYTESUM(tab[FAST_SCAN_AT(k)].x);
            u32 wt;
            u32 e = wave_excl_scan(sum, wt);
            if (lane == 63) tot[wv] = wt;
            __syncthreads();                                                        // (3)
            u32 wb = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) if (k < (int)wv) wb += tot[k];
            e += wb;
#pragma unroll
            for (int k = 0; k < EW; ++k) { tab[FAST_SCAN_AT(k)].y = e; e += BYTESUM(tab[FAST_SCAN_AT(k)].x); }
            __syncthreads();                                                        // (4)
            F2P(3);
        }
        F2_LOAD(3 * LB, 4 * LB);
        const bool early_exit = !ok;                             // the rest is requested behind the block below (not
                                                                 // here: loads pending on one path make the register
                                                                 // reuse of the other path wait for them)
        if (ok) {                                                // ---- phase 4: every row fetches its sub-bucket
            u32 bc[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                bc[j] = 0;
                if (j < rows) {
                    const u32 c = key[j];
                    const u32 dg = c != 0xffffffffu ? (c >> (6 + sh)) : 0u, by = (dg & 3u) * 8u;
                    const uint2 e = tab[FAST_W(dg >> 2)];
                    bc[j] = (e.y + BYTESUM(e.x & ((1u << by) - 1u))) | (((e.x >> by) & 255u) << 16);
                }
            }
            __syncthreads();                                                        // (5) the table is dead: composites move in
            F2P(4);
            F2_LOAD(4 * LB, 5 * LB);
            // ---- phase 5: composites to their slots
            // key[] := sub-bucket base | size << 16 | my arrival rank << 24 ; the composite itself goes to LDS and is
            // read back by its owner (one register per row instead of two while the next segment's records are in flight)
            if (t < 64) ex[len + t] = 0xffffffffu;               // inert tail for the probes
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) {
                    const u32 c = key[j];
                    const bool v = c != 0xffffffffu;
                    ex[v ? (bc[j] & 0xffffu) + (c & 63u) : TRASH_POS] = c;
                    key[j] = v ? (bc[j] | ((c & 63u) << 24)) : TRASH_POS;
                }
            __syncthreads();                                                        // (6)
            F2P(5);
            F2_LOAD(5 * LB, NL);
            // ---- phase 6: final rows, exchange, output
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (j < rows) {
                    const u32 pk = key[j], b0 = pk & 0xffffu, bsz = (pk >> 16) & 255u;
                    const u32 me = ex[b0 + (pk >> 24)];
                    u32 lt = 0, eq = 0;
#pragma unroll
                    for (int q = 0; q < FAST_PROBE; ++q) { const u32 c = ex[b0 + q]; lt += c < me; eq += (c ^ me) < 64u; }
                    const bool v = me != 0xffffffffu;
                    const bool slow = v && (bsz > FAST_PROBE || eq > 1);
                    key[j] = v ? b0 + lt : TRASH_POS;            // final row; bit 31: tied (handled through the tie list)
                    if (__ballot(slow)) {
                        if (slow) {      // rare: long sub-bucket or equal keys
                            const u32 b1 = b0 + bsz;
                            u32 ltk = 0;
                            lt = 0; eq = 0;
#pragma nounroll
                            for (u32 q = b0; q < b1; ++q) { const u32 c = ex[q]; lt += c < me; eq += (c ^ me) < 64u; ltk += (c >> 6) < (me >> 6); }
                            key[j] = b0 + lt;
                            if (eq > 1) {
                                const u32 ro = lt - ltk;
                                const u32 slot = atomicAdd(&misc[4], 1u);
                                u32 loc = 0;
                                if (ro == 0) loc = atomicAdd(&misc[eq <= TINY_MAX ? 2 : 3], eq);
                                if (slot < (u32)TL) { tl[3 * slot] = exi[FAST_P(j)]; tl[3 * slot + 1] = (b0 + ltk) | (eq << 16) | (ro << 24); tl[3 * slot + 2] = loc; }
                                key[j] |= 0x80000000u;
                            }
                        }
                    }
                }
            }
            __syncthreads();                                                        // (7) probing is over: ex becomes the exchange
            F2P(6);
            // room for the tied runs: the two returning atomics are issued now and looked at after the exchange
            if (t == 0 && misc[4] != 0 && misc[4] <= (u32)TL) {
                if (misc[2]) res_t = atomicAdd(&counters[em.pool_cnt_idx], misc[2]);
                if (misc[3]) res_s = atomicAdd(&counters[em.seg_cnt_idx], misc[3]);
            }
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) {
                    const u32 p = FAST_P(j);
                    if (p < len) {
                        const u32 id = exi[p];
                        ex[key[j] & 0xffffu] = id;
                        if (mode == MODE_ISA && !(key[j] >> 31) && (key[j] != 0 || stale)) isa[id] = rank0 + sa_off + key[j] + 1u;
                    }
                }
            __syncthreads();                                                        // (8)
            F2P(7);
            const u32 nt = misc[4];
            if (t == 0 && nt != 0 && nt <= (u32)TL) {
                u32 bad = 0;
                if ((u64)res_t + misc[2] > em.pool_cap) { atomicOr(&counters[C_ERR], 32u); bad = 1; }
                if ((u64)res_s + misc[3] > em.seg_cap) { atomicOr(&counters[C_ERR], 64u); bad = 1; }
                misc[6] = res_t; misc[7] = res_s; misc[8] = bad;
            }
            {   // rows out, 8 bytes per lane at 8-byte aligned addresses: lane pairs start at an even ROW of the array
                const u32 shift = sa_off & 1u;
                u32* outp = sa_out + sa_off;
                if (shift && t == 0) outp[0] = ex[0];
#pragma unroll
                for (int q = 0; q < NL; ++q)
                    if (2 * q < rows) {
                        const u32 p = FAST_P(2 * q) + shift;
                        if (p + 1 < len) {
                            uint2 v; v.x = ex[p]; v.y = ex[p + 1];
                            *reinterpret_cast<uint2*>(outp + p) = v;
                        } else if (p < len) outp[p] = ex[p];
                    }
            }
            if (nt > (u32)TL) ok = false;                       // too many ties for the list: k_sort_mid redoes the segment
            else if (nt) {                                       // block-uniform
                __syncthreads();                                 // rows are out: ex[run start] now carries the run's local offset
                for (u32 i = t; i < nt; i += THREADS) { const u32 w1 = tl[3 * i + 1]; if ((w1 >> 24) == 0) ex[w1 & 0xffffu] = tl[3 * i + 2]; }
                __syncthreads();
                if (!misc[8]) {
                    const u32 base_t = misc[6], base_s = misc[7];
                    for (u32 i = t; i < nt; i += THREADS) {
                        const u32 id = tl[3 * i], w1 = tl[3 * i + 1];
                        const u32 rs = w1 & 0xffffu, rl = (w1 >> 16) & 255u, ro = w1 >> 24;
                        if (mode == MODE_ISA && (rs != 0 || stale)) isa[id] = rank0 + sa_off + rs + 1u;
                        if (rl <= TINY_MAX) {
                            const u32 o = base_t + ex[rs] + ro;
                            em.pool_rec[o] = (u64)id;
                            em.pool_hdr[o] = pack_hdr(sa_off + rs, rl, ro);
                        } else {
                            const u32 o = base_s + ex[rs] + ro;
                            em.seg_rec[o] = (u64)id;
                            if (ro == 0) { const Desc nd = {o, rl, sa_off + rs, em.seg_buf}; push_desc(em.lists, counters, class_of(rl), nd); }
                        }
                    }
                }
            }
        }
        F2P(8);
        if (early_exit) F2_LOAD(4 * LB, NL);
        if (!ok && len != 0 && t == 0) fb_list[atomicAdd(&counters[fb_cnt_idx], 1u)] = cur;   // leave it to k_sort_mid
        if (!more) break;
        __syncthreads();            // everyone is done with this segment's LDS before it is reset
        F2P(9);
    }
#ifdef FAST2_PROF
    if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_fast2_prof[i], prof_acc[i]);
#endif
#undef FAST_P
#undef FAST_W
#undef FAST_SCAN_AT
#undef BYTESUM
#undef FAST_SRC
#undef F2_LOAD
}

template <int THREADS, int ITEMS, int BITS, int TL>
constexpr size_t sort_fast2_lds_bytes()
{
    return ((size_t)THREADS * ITEMS * 2 + 64 + 3 * (size_t)TL + 16 + 16) * 4;
}

template <int THREADS, int ITEMS, int BITS>
constexpr size_t sort_fast_lds_bytes()
{
    return ((size_t)THREADS * ITEMS + 64 + ((size_t)1 << BITS) + 16 + 16 + 16) * 4;
}

// persistent launch: workgroups stride over the list and skip what k_sort_fast already finished
// (the class-B instance asks for 4 waves per SIMD = 128 VGPRs: one more resident workgroup per CU)
template <int THREADS, int ITEMS, bool W, bool AUX = false>
__global__ __launch_bounds__(THREADS, (THREADS == 256 ? 4 : 1)) void k_sort_mid(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                      typename Wd<W>::sa_t* __restrict__ sa_out, u32* __restrict__ isa, u32 mode,
                                                      Emit em, u32* __restrict__ counters, const u32* __restrict__ ids, u32 ids_cnt_idx,
                                                      GatherSpec g, const u8* __restrict__ code)
{
    // ids != nullptr: only the segments k_sort_fast left behind (their count lives in the counters block)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_k[];
    __shared__ u8 s_code[256];
    u32* ach = reinterpret_cast<u32*>(smem_k + sort_mid_lds_bytes<THREADS, ITEMS>()) - 24;
    if (threadIdx.x < 24) ach[threadIdx.x] = 0;
    if (g.text) for (u32 i = threadIdx.x; i < 256u; i += THREADS) s_code[i] = code[i];
    __syncthreads();
#ifdef MID_PROF
    __shared__ unsigned long long mprof[16];
    if (threadIdx.x == 0) { for (int i = 0; i < 15; ++i) mprof[i] = 0; mprof[15] = clock64(); }
#endif
    const u32 total = ids ? counters[ids_cnt_idx] : nseg;
    if (blockIdx.x < total) {
        // the next descriptor is fetched while the current segment is sorted (one memory round trip less per segment)
        Desc d = list[ids ? ids[blockIdx.x] : blockIdx.x];
        for (u32 i = blockIdx.x; i < total; i += gridDim.x) {
            const u32 ni = i + gridDim.x < total ? i + gridDim.x : i;
            const Desc dn = list[ids ? ids[ni] : ni];
            sort_mid_segment<THREADS, ITEMS, W, AUX>(bufs, d, sa_out, isa, mode, em, counters, g, s_code MPROF_ARG);
            __syncthreads();
            d = dn;
        }
    }
#ifdef MID_PROF
    if (threadIdx.x == 0) for (int i = 0; i < 15; ++i) atomicAdd(&g_mid_prof[THREADS == 64 ? 0 : (THREADS == 256 ? 1 : 2)][i], mprof[i]);
#endif
    // give back what is left of my chunks as neutral entries
    for (u32 i = ach[0] + threadIdx.x; i < ach[1]; i += THREADS) { em.pool_rec[i] = 0; em.pool_hdr[i] = 0; }
    for (u32 i = ach[2] + threadIdx.x; i < ach[3]; i += THREADS) em.seg_rec[i] = 0;
#pragma unroll
    for (u32 k = 0; k < 3; ++k)
        for (u32 i = ach[4 + 2 * k] + threadIdx.x; i < ach[5 + 2 * k]; i += THREADS) { const Desc z = {0, 0, 0, 0}; em.lists.cls[k][i] = z; }
}


// ------------------------------------------------------------------------------------------------
// Class-A segments (33 .. 512 records), SEVERAL PER WAVE (round 6; the role of multikey_insertion_sort / the small
// partitions of multikey_quicksort, reference cpp:223-312, cpp:488-642).
// The one-wave instance of k_sort_mid handles one segment at a time: a chain of four or five dependent memory round trips
// (descriptor -> records -> keys -> reservation -> emitted records) per ~100 records, ~46 K cycles whatever the length
// (profiles/r05_mid_prof_text.txt) - the largest single kernel of the text build.  Here a wave takes a BUNDLE of 16
// consecutive descriptors (one coalesced 256-byte load, the next bundle's in flight while this one is sorted) and packs
// them greedily into TILES of up to 8 segments / 512 records; a tile goes through every phase ONCE:
//   * all its records are loaded (and, in text rounds, all its keys gathered) in one round trip;
//   * ONE least-significant-digit sort of the tile on the virtual key  segment number : key bits that vary inside some
//     segment  (bits above the highest varying bit are constant inside every segment, so the segment number takes their
//     place: text keys of 29 bits + 3 bits of segment number are the same four 8-bit passes one segment needs);
//     what is permuted per pass is (key, slot): the slot word names the record's place in the loaded tile (and its
//     segment), the suffix index and the companion are fetched through LDS once, after the last pass;
//   * equal-key runs from ballot bitmaps as in sort_mid_segment, with a run boundary forced at every segment boundary;
//   * ONE reservation and ONE compaction of the still-tied runs for the whole tile.
// Rows, ranks, group heads, emitted records and descriptors are what sort_mid_segment writes for the same segments.
// ------------------------------------------------------------------------------------------------
#define MIDT_BUN 16          // descriptors per bundle
#define MIDT_KMAX 8          // segments per tile
#ifndef MIDT_GATHER_B
#define MIDT_GATHER_B 8      // key gathers of a lane in flight together
#endif
#ifndef MIDT_WAVES
#define MIDT_WAVES 4         // waves per SIMD the register allocation aims at (128 VGPRs)
#endif
template <bool W, bool AUX = false>
__global__ __launch_bounds__(64, MIDT_WAVES) void k_sort_mid_tiles(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                      typename Wd<W>::sa_t* __restrict__ sa_out, u32* __restrict__ isa, u32 mode,
                                                      Emit em, u32* __restrict__ counters, GatherSpec g, const u8* __restrict__ code)
{
    constexpr u32 KL = klow<W>();
    constexpr int ITEMS = 8, CAP = 512, NW = 8;
    static_assert(CAP_A <= CAP, "a class-A segment must fit one tile");
    __shared__ __attribute__((aligned(16))) u64 ex2[CAP];      // LSD passes: (key, slot) pairs travel as one 8-byte word
    u32* const ex = reinterpret_cast<u32*>(ex2);                 // ... everything else uses the first CAP 32-bit words
    __shared__ u32 wcnt[256];
    __shared__ u64 bm_eq[NW], bm_tiny[NW], bm_seg[NW];
    __shared__ u32 pre_tiny[NW], pre_seg[NW];
    __shared__ u32 misc[8], ach[24];
    __shared__ __attribute__((aligned(16))) Desc dtab[MIDT_BUN];
    __shared__ u32 st_pre[MIDT_KMAX + 1], st_sab[MIDT_KMAX], st_flag[MIDT_KMAX], st_key0[MIDT_KMAX];
    __shared__ u32 st_roff[MIDT_KMAX];                        // first record of the segment (in its record buffer) minus its first tile position
    __shared__ u8 s_code[256];
#ifdef MID_PROF
    __shared__ unsigned long long mprof[16];
    if (threadIdx.x == 0) { for (int i = 0; i < 15; ++i) mprof[i] = 0; mprof[15] = clock64(); }
#endif
    const u32 lane = threadIdx.x;
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (lane < 24) ach[lane] = 0;
    if (g.text) for (u32 i = lane; i < 256u; i += 64u) s_code[i] = code[i];
    const u32 nbun = (nseg + MIDT_BUN - 1) / MIDT_BUN;
    const u32 rank0 = counters[C_RANK0];
    // (the descriptor in flight is held as four plain words: a struct assigned under a condition lands in scratch memory, with a wait
    // for the load right behind it)
    const uint4* list4 = reinterpret_cast<const uint4*>(list);
    u32 dn0 = 0, dn1 = 0, dn2 = 0, dn3 = 0;
    { const u64 i = (u64)blockIdx.x * MIDT_BUN + lane; if (lane < MIDT_BUN && i < nseg) { const uint4 v = list4[i]; dn0 = v.x; dn1 = v.y; dn2 = v.z; dn3 = v.w; } }
    for (u32 b = blockIdx.x; b < nbun; b += gridDim.x) {
        __syncthreads();
        if (lane < MIDT_BUN) *reinterpret_cast<uint4*>(&dtab[lane]) = make_uint4(dn0, dn1, dn2, dn3);
        {   // the next bundle's descriptors travel while this one is sorted
            const u64 i = ((u64)b + gridDim.x) * MIDT_BUN + lane;
            dn0 = dn1 = dn2 = dn3 = 0;
            if (lane < MIDT_BUN && i < nseg) { const uint4 v = list4[i]; dn0 = v.x; dn1 = v.y; dn2 = v.z; dn3 = v.w; }
        }
        __syncthreads();
        u32 start = 0;
        while (start < MIDT_BUN) {
            // ---- pack the next tile: consecutive descriptors while they fit (neutral entries, len 0, are skipped) ----
            u32 e = start, T = 0, K = 0;
            while (e < MIDT_BUN) {
                const u32 l = dtab[e].len;
                if (l != 0) {
                    if (K == 0u && l > (u32)CAP) { if (lane == 0) atomicOr(&counters[C_ERR], 0x800u); ++e; break; }      // (not a class-A segment: cannot happen; reported, not looped on)
                    if (K == MIDT_KMAX || T + l > (u32)CAP) break;
                    if (lane == 0) {
                        const Desc d = dtab[e];
                        st_pre[K] = T; st_sab[K] = d.sa_off - T; st_flag[K] = d.buf; st_roff[K] = d.rec_off - T;
                    }
                    T += l; ++K;
                }
                ++e;
            }
            start = e;
            if (K == 0) continue;          // (nothing but neutral entries up to `start`: the loop ends when start reaches the bundle's end)
            if (lane <= MIDT_KMAX && lane >= K) st_pre[lane] = lane == K ? T : 0xffffffffu;
            if (lane < 8) misc[lane] = 0;
            __syncthreads();
            MPROF(6);
#ifdef MID_PROF
            if (threadIdx.x == 0) { mprof[8] += 1; mprof[9] += T; }
#endif
            const int rows = (int)((T + 63u) / 64u);
            // segment of every tile position this lane holds (3 bits per slot; the same before and after the sort)
            u32 segpack = 0;
            {
                u32 pk[MIDT_KMAX - 1];
#pragma unroll
                for (int k = 0; k < MIDT_KMAX - 1; ++k) pk[k] = st_pre[k + 1];
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const u32 p = j * 64 + lane;
                    u32 sg = 0;
#pragma unroll
                    for (int k = 0; k < MIDT_KMAX - 1; ++k) sg += p >= pk[k] ? 1u : 0u;
                    segpack |= sg << (3 * j);
                }
            }
#define MIDT_SEG(j) ((segpack >> (3 * (j))) & 7u)
            u32 key[ITEMS], idx[ITEMS], slot[ITEMS], pos[ITEMS];
            u32 ax[AUX ? ITEMS : 1];
            // ---- all records of the tile in one round trip ----
            if (g.text == nullptr) {
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const u32 p = j * 64 + lane;
                    key[j] = 0xffffffffu; idx[j] = 0xffffffffu; slot[j] = 0xffffffffu;
                    if constexpr (AUX) ax[j] = 0;
                    if (j < rows && p < T) {
                        const u32 sg = MIDT_SEG(j);
                        const u32 bi = st_flag[sg] & 3u, ro = st_roff[sg] + p;
                        const u64 r = (bi == 0u ? bufs.p[0] : (bi == 1u ? bufs.p[1] : bufs.p[2]))[ro];
                        key[j] = (u32)(r >> 32); idx[j] = (u32)r; slot[j] = p | (sg << 16);
                        if constexpr (AUX) ax[j] = (bi == 0u ? bufs.x[0] : (bi == 1u ? bufs.x[1] : bufs.x[2]))[ro];
                    }
                }
            } else {        // text round: the records carry the suffix index only, the keys are gathered here
                typename Wd<W>::sa_t fi[ITEMS];
                bool valid[ITEMS];
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const u32 p = j * 64 + lane;
                    valid[j] = j < rows && p < T;
                    const u32 sg = MIDT_SEG(j);
                    const u32 bi = st_flag[sg] & 3u, ro = st_roff[sg] + p;
                    fi[j] = valid[j] ? rec_idx<W>((bi == 0u ? bufs.p[0] : (bi == 1u ? bufs.p[1] : bufs.p[2]))[ro]) : 0;
                    slot[j] = valid[j] ? (p | (sg << 16)) : 0xffffffffu;
                }
#ifdef MID_PROF
                asm volatile("" :: "v"(fi[0]));
                MPROF(0);
#endif
                if constexpr (AUX) {        // (gather rounds of a two-stage build: the companion slot carries the characters in front of the suffix)
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) ax[j] = (valid[j] && g.pc_out) ? pc_fetch(g.text, (u32)fi[j]) : PC_UNKNOWN;
                }
                gather_keys<W, ITEMS, MIDT_GATHER_B>(g, s_code, fi, valid, key);
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    idx[j] = valid[j] ? (u32)fi[j] : 0xffffffffu;
                    if constexpr (W) { if (valid[j]) key[j] = (key[j] & 0xffffff00u) | (u32)(fi[j] >> 32); }      // the index byte rides in the key word
                }
            }
            // ---- key bits that vary inside some segment (against each segment's first key) ----
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 p = j * 64 + lane;
                if (j < rows && p < T) { const u32 sg = MIDT_SEG(j); if (p == st_pre[sg]) st_key0[sg] = key[j]; }
            }
            __syncthreads();
            u32 diff = 0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 p = j * 64 + lane;
                if (j < rows && p < T) diff |= key[j] ^ st_key0[MIDT_SEG(j)];
            }
            diff = wave_or(diff) & (0xffffffffu << KL);      // (wide: the index byte never decides a pass; equal keys keep their order)
            MPROF(1);
            if (diff) {
                // virtual key = segment : key bits [tz, hb]; the records start out in segment order, so without varying bits nothing moves
                const u32 tz = (u32)__ffs((int)diff) - 1u, hb = 31u - (u32)__clz((int)diff);
                const u32 nbits = hb - tz + 1u;
                const u32 sbits = K > 1u ? 32u - (u32)__clz((int)(K - 1u)) : 0u;
                const u32 kmask = nbits >= 32u ? 0xffffffffu : ((1u << nbits) - 1u);
                const u32 dvar = diff >> tz;
#pragma nounroll
                for (u32 sh = 0; sh < nbits + sbits; sh += 8) {
                    if (sh + 8u <= nbits && ((dvar >> sh) & 255u) == 0u) continue;      // byte equal inside every segment, no segment bits in it
                    for (u32 i = lane; i < 256u; i += 64u) wcnt[i] = 0;
                    __syncthreads();
#define MIDT_DIGIT(j) ((u32)(((((u64)(slot[j] >> 16)) << nbits) | (u64)((key[j] >> tz) & kmask)) >> sh) & 255u)
                    if (!em.safe_rank) {
                        // stable rank inside the wave = what a returning LDS atomic hands back (see sort_mid_segment)
#pragma unroll
                        for (int j = 0; j < ITEMS; ++j)
                            if (j < rows) { const u32 dg = MIDT_DIGIT(j); pos[j] = atomicAdd(&wcnt[dg], 1u) | (dg << 16); }
                    } else {
#pragma unroll
                        for (int j = 0; j < ITEMS; ++j) {
                            if (j < rows) {
                                const u32 dg = MIDT_DIGIT(j);
                                u64 mask = ~0ull;
#pragma unroll
                                for (int bb = 0; bb < 8; ++bb) {
                                    const bool bit = (dg >> bb) & 1u;
                                    const u64 bal = __ballot(bit);
                                    mask &= bit ? bal : ~bal;
                                }
                                const int leader = __ffsll((long long)mask) - 1;
                                u32 old = 0;
                                if ((int)lane == leader) old = atomicAdd(&wcnt[dg], (u32)__popcll(mask));
                                old = __shfl(old, leader, 64);
                                pos[j] = (old + (u32)__popcll(mask & lt_mask)) | (dg << 16);
                            }
                        }
                    }
#undef MIDT_DIGIT
                    __syncthreads();
                    scan256_first_wave(wcnt, wcnt);      // in place: a lane reads its four counts, then writes their prefixes
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j)
                        if (j < rows) { pos[j] = (pos[j] & 0xffffu) + wcnt[pos[j] >> 16]; ex2[pos[j]] = ((u64)slot[j] << 32) | key[j]; }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) if (j < rows) { const u64 v = ex2[j * 64 + lane]; key[j] = (u32)v; slot[j] = (u32)(v >> 32); }
                    __syncthreads();
                }
                // the suffix indices (and companions) follow their records: one exchange, whatever the number of passes
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) if (j < rows) ex[j * 64 + lane] = idx[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) if (j < rows) idx[j] = slot[j] != 0xffffffffu ? ex[slot[j] & 0xffffu] : 0xffffffffu;
                __syncthreads();
                if constexpr (AUX) {
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) if (j < rows) ex[j * 64 + lane] = ax[j];
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) if (j < rows) ax[j] = slot[j] != 0xffffffffu ? ex[slot[j] & 0xffffu] : 0u;
                    __syncthreads();
                }
            }
            MPROF(2);
            // ---- equal-key runs (never across a segment boundary) ----
            u32 rs[ITEMS], rl[ITEMS];
            bool any_eq = false;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) if (j < rows) ex[j * 64 + lane] = key[j];
            if (lane < (u32)NW) { bm_eq[lane] = 0; bm_tiny[lane] = 0; bm_seg[lane] = 0; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) {
                    const u32 p = j * 64 + lane;
                    const bool inner = (p < T) && (p + 1u != st_pre[MIDT_SEG(j) + 1u]);        // my right neighbour belongs to my segment
                    const u32 nk = ex[p + 1u < (u32)CAP ? p + 1u : (u32)CAP - 1u] >> KL;
                    const bool eqn = inner && ((key[j] >> KL) == nk);
                    if (inner && ((key[j] >> KL) > nk)) atomicOr(&counters[C_ERR], 0x200u);      // not sorted: the LDS-atomic ranks were not stable after all
                    const u64 bal = __ballot(eqn);
                    if (lane == 0) bm_eq[j] = bal;
                    any_eq |= (bal != 0);
                }
            __syncthreads();
            {   // pre_tiny[w] = start of a run that enters word w from the left, pre_seg[w] = end of a run that leaves it to the right
                const int w = (int)lane;
                const u64 z = w < NW ? ~bm_eq[w] : ~0ull;
                const u32 vs = z ? (u32)((w << 6) + 64 - __clzll((long long)z)) : 0u;
                const u32 ve = z ? (u32)((w << 6) + __ffsll((long long)z) - 1) : 0xffffffffu;
                const u32 pmx = wave_incl_scan_max_dpp(vs);
                const u32 smn = ~(u32)__shfl(wave_incl_scan_max_dpp(~(u32)__shfl(ve, 63 - (int)lane, 64)), 63 - (int)lane, 64);
                u32 run_s = __shfl_up(pmx, 1, 64); if (lane == 0) run_s = 0;
                u32 run_e = __shfl_down(smn, 1, 64); if (lane == 63) run_e = 0xffffffffu;
                if (w < NW) { pre_tiny[w] = run_s; pre_seg[w] = run_e; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) {
                    const u32 p = j * 64 + lane;
                    rs[j] = p; rl[j] = 1;
                    if (p < T) {
                        const u32 w = p >> 6, bit = p & 63u;
                        const u64 z = ~bm_eq[w];
                        const u64 below = bit ? (z & (~0ull >> (64 - bit))) : 0ull;
                        const u64 above = z & (~0ull << bit);
                        const u32 s = below ? (u32)((w << 6) + 64 - __clzll((long long)below)) : pre_tiny[w];
                        const u32 e2 = above ? (u32)((w << 6) + __ffsll((long long)above) - 1) : pre_seg[w];
                        rs[j] = s; rl[j] = e2 - s + 1;
                        const u32 sg = MIDT_SEG(j);
                        const u32 sab = st_sab[sg];                     // the segment's first row minus its first tile position
                        sa_out[sab + p] = full_idx<W>(key[j], idx[j]);
                        if constexpr (AUX) { if (g.text && g.pc_out && rl[j] == 1u) g.pc_out[sab + p] = ax[j]; }
                        if (mode == MODE_ISA && (s != st_pre[sg] || (st_flag[sg] & DESC_STALE))) isa[idx[j]] = rank0 + sab + s + 1u;
                        if (mode == MODE_DEFER) em.grp_out[sab + p] = sab + s;
                    }
                }
            MPROF(3);
            if (em.discard || !any_eq) continue;           // (any_eq is wave-uniform: it comes from ballots)

            // ---- compact the still-tied runs of the whole tile into next round's structures: ONE reservation ----
            u32 hc[3] = {0, 0, 0};           // run heads per size class (wave-uniform)
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) {
                    const u32 p = j * 64 + lane;
                    const bool tied = (p < T) && rl[j] > 1;
                    const u64 bt = __ballot(tied && rl[j] <= TINY_MAX);
                    const u64 bs = __ballot(tied && rl[j] > TINY_MAX);
                    if (lane == 0) { bm_tiny[j] = bt; bm_seg[j] = bs; }
                    if (bs) {
                        const bool head = tied && rl[j] > TINY_MAX && p == rs[j];
                        const u32 cls = class_of(rl[j]);
#pragma unroll
                        for (u32 k = 0; k < 3; ++k) hc[k] += (u32)__popcll(__ballot(head && cls == k));
                    }
                }
            __syncthreads();
            {
                const int w = (int)lane;
                const u32 ct = w < NW ? (u32)__popcll(bm_tiny[w]) : 0u;
                const u32 cs = w < NW ? (u32)__popcll(bm_seg[w]) : 0u;
                u32 tt, ts;
                const u32 et = wave_excl_scan(ct, tt);
                const u32 es = wave_excl_scan(cs, ts);
                if (w < NW) { pre_tiny[w] = et; pre_seg[w] = es; }
                if (lane == 0) {
                    u32 bt = 0, bs = 0, bad = 0;
                    u32* pf = ach + 10;                       // pending fills: (begin, end) x {pool, seg, desc A, desc B, desc C}
#pragma unroll
                    for (int k = 0; k < 10; ++k) pf[k] = 0;
                    if (tt) {
                        if (ach[0] + tt > ach[1]) {
                            pf[0] = ach[0]; pf[1] = ach[1];
                            const u32 sz = tt > em.pool_chunk ? tt : em.pool_chunk;
                            const u32 b0 = atomicAdd(&counters[em.pool_cnt_idx], sz);
                            if ((u64)b0 + sz > em.pool_cap) { atomicOr(&counters[C_ERR], 32u); bad = 1; ach[0] = 0; ach[1] = 0; }
                            else { ach[0] = b0; ach[1] = b0 + sz; }
                        }
                        if (!bad) { bt = ach[0]; ach[0] += tt; }
                    }
                    if (ts) {
                        if (ach[2] + ts > ach[3]) {
                            pf[2] = ach[2]; pf[3] = ach[3];
                            const u32 sz = ts > em.seg_chunk ? ts : em.seg_chunk;
                            const u32 b0 = atomicAdd(&counters[em.seg_cnt_idx], sz);
                            if ((u64)b0 + sz > em.seg_cap) { atomicOr(&counters[C_ERR], 64u); bad = 1; ach[2] = 0; ach[3] = 0; }
                            else { ach[2] = b0; ach[3] = b0 + sz; }
                        }
                        if (!bad) { bs = ach[2]; ach[2] += ts; }
#pragma unroll
                        for (u32 k = 0; k < 3; ++k) {
                            const u32 need = hc[k];
                            if (need && ach[4 + 2 * k] + need > ach[5 + 2 * k]) {
                                pf[4 + 2 * k] = ach[4 + 2 * k]; pf[5 + 2 * k] = ach[5 + 2 * k];
                                const u32 dch = em.seg_chunk ? (k == 0 ? 64u : (k == 1 ? 16u : 4u)) : 0u;      // (chunk per DESTINATION class: sort_mid_segment)
                                const u32 sz = need > dch ? need : dch;
                                const u32 b0 = atomicAdd(&counters[em.lists.cnt_idx + k], sz);
                                if ((u64)b0 + sz > em.lists.cap[k]) { atomicOr(&counters[C_ERR], 1u); bad = 1; ach[4 + 2 * k] = 0; ach[5 + 2 * k] = 0; }
                                else { ach[4 + 2 * k] = b0; ach[5 + 2 * k] = b0 + sz; }
                            }
                        }
                    }
                    misc[1] = bt; misc[2] = bs; misc[3] = bad;
                }
            }
            __syncthreads();
            MPROF(4);
            if (misc[3]) continue;
            {   // neutral-fill the chunk tails that were just abandoned
                const u32* pf = ach + 10;
                for (u32 i = pf[0] + lane; i < pf[1]; i += 64u) { em.pool_rec[i] = 0; em.pool_hdr[i] = 0; }
                for (u32 i = pf[2] + lane; i < pf[3]; i += 64u) em.seg_rec[i] = 0;
#pragma unroll
                for (u32 k = 0; k < 3; ++k)
                    for (u32 i = pf[4 + 2 * k] + lane; i < pf[5 + 2 * k]; i += 64u) { const Desc z = {0, 0, 0, 0}; em.lists.cls[k][i] = z; }
            }
            const u32 base_t = misc[1], base_s = misc[2];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (j < rows) {
                    const u32 p = j * 64 + lane;
                    if (p < T && rl[j] > 1) {
                        const u32 sab = st_sab[MIDT_SEG(j)];
                        if (rl[j] <= TINY_MAX) {
                            const u32 o = base_t + pre_tiny[j] + (u32)__popcll(bm_tiny[j] & lt_mask);
                            u64 r = (u64)full_idx<W>(key[j], idx[j]);
                            if constexpr (AUX) { if (!g.text) r |= (u64)ax[j] << 32; }        // next round's key comes with the record
                            em.pool_rec[o] = r;
                            em.pool_hdr[o] = pack_hdr(sab + rs[j], rl[j], p - rs[j]);
                        } else {
                            const u32 o = base_s + pre_seg[j] + (u32)__popcll(bm_seg[j] & lt_mask);
                            u64 r = (u64)full_idx<W>(key[j], idx[j]);
                            if constexpr (AUX) { if (!g.text) r |= (u64)ax[j] << 32; }
                            em.seg_rec[o] = r;
                            if (p == rs[j]) {
                                const Desc nd = {o, rl[j], sab + rs[j], em.seg_buf};
                                const u32 cls = class_of(rl[j]);
                                em.lists.cls[cls][atomicAdd(&ach[4 + 2 * cls], 1u)] = nd;      // room was reserved above
                            }
                        }
                    }
                }
            MPROF(5);
#undef MIDT_SEG
            __syncthreads();
        }
    }
#ifdef MID_PROF
    if (threadIdx.x == 0) for (int i = 0; i < 15; ++i) atomicAdd(&g_mid_prof[0][i], mprof[i]);
#endif
    __syncthreads();
    // give back what is left of my chunks as neutral entries
    for (u32 i = ach[0] + lane; i < ach[1]; i += 64u) { em.pool_rec[i] = 0; em.pool_hdr[i] = 0; }
    for (u32 i = ach[2] + lane; i < ach[3]; i += 64u) em.seg_rec[i] = 0;
#pragma unroll
    for (u32 k = 0; k < 3; ++k)
        for (u32 i = ach[4 + 2 * k] + lane; i < ach[5 + 2 * k]; i += 64u) { const Desc z = {0, 0, 0, 0}; em.lists.cls[k][i] = z; }
}


// length of the common prefix of the suffixes a and b (8 bytes per step)
__device__ __forceinline__ u32 dev_match_length(const u8* __restrict__ text, u64 n, u64 a, u64 b)
{
    if (a > b) { const u64 x = a; a = b; b = x; }
    u64 m = 0;
    while (b + m + 8 <= n) {
        u64 x, y;
        __builtin_memcpy(&x, text + a + m, 8);
        __builtin_memcpy(&y, text + b + m, 8);
        if (x != y) { m += (u64)(__ffsll((long long)(x ^ y)) - 1) >> 3; return (u32)m; }
        m += 8;
    }
    while (b + m < n && text[a + m] == text[b + m]) ++m;
    return (u32)m;
}

// Exact order of two suffixes that are known to agree on their first `depth` characters (zero-padded view): true if suffix
// `other` is smaller than suffix `mine`.  The suffix that ends first is the smaller one (the reference's implicit sentinel).
// A comparison that would run past `cap` further bytes gives up (flag 0x400: the caller's build is abandoned).
__device__ __forceinline__ bool deep_other_less(const u8* __restrict__ text, u64 n, u32 mine, u32 other, u64 depth, u32 cap, u32* __restrict__ counters)
{
    u64 a = mine, b = other;
    const bool swapped = a > b;
    if (swapped) { const u64 x = a; a = b; b = x; }            // a < b: b is the shorter suffix
    bool b_less;
    if (b + depth >= n) b_less = true;                          // b ends inside the part already known to agree
    else {
        u64 m = depth;
        bool done = false;
        while (b + m + 8 <= n) {
            u64 x, y;
            __builtin_memcpy(&x, text + a + m, 8);
            __builtin_memcpy(&y, text + b + m, 8);
            if (x != y) { m += (u64)(__ffsll((long long)(x ^ y)) - 1) >> 3; done = true; break; }
            m += 8;
            if (m - depth > cap) { atomicOr(&counters[C_ERR], 0x400u); done = true; break; }
        }
        if (!done) while (b + m < n && text[a + m] == text[b + m]) ++m;
        b_less = (b + m >= n) || text[b + m] < text[a + m];
    }
    return swapped ? !b_less : b_less;       // b is `other` unless swapped
}

// ------------------------------------------------------------------------------------------------
// Tiny runs (2..32 records), millions of them on text/DNA inputs: flat pool, one lane per record,
// rank by counting inside the run (the insertion-sort regime of cpp:223-312).  A workgroup owns the
// runs that START in its 256-slot window and loads a 31-slot halo.
// ------------------------------------------------------------------------------------------------
template <bool W>
__global__ __launch_bounds__(256) void k_sort_tiny(const u64* __restrict__ pool_rec, const u64* __restrict__ pool_hdr,
                                                   u32 cnt_idx, typename Wd<W>::sa_t* __restrict__ sa_out, u32* __restrict__ isa, u32 mode,
                                                   u64* __restrict__ next_rec, u64* __restrict__ next_hdr, u32 next_cnt_idx, u32 next_cap,
                                                   u32 chunk, u32* __restrict__ counters, u32* __restrict__ grp_out, u32 discard,
                                                   GatherSpec g, const u8* __restrict__ code, u32 deep_cap,
                                                   const u32* __restrict__ pool_aux = nullptr /* round 0, small alphabets: companions of the pool records (RecBufs::x) */)
{
    // deep_cap != 0 (narrow text rounds of a two-stage build, once the pool is small): the runs are finished HERE by comparing
    // the suffixes themselves, however long they agree (repeated passages: thousands of key rounds otherwise) - nothing goes
    // to the next round
    constexpr u32 KL = klow<W>();
    const bool deep = !W && deep_cap != 0u && g.text != nullptr;
    __shared__ u8 s_code[256];
    if (g.text) s_code[threadIdx.x] = code[threadIdx.x];
    constexpr int WIN = 256, HALO = TINY_MAX, TOT = WIN + HALO;
    __shared__ u32 lkey[TOT], lrun[TOT], lkey2[TOT];
    __shared__ u32 s_total, s_base, s_cb, s_ce, s_fb, s_fe;      // window total / base; chunk [cb, ce); pending fill [fb, fe)
    const bool dbl = !W && !deep && g.text != nullptr && (g.flags & GS_TINY2) != 0u;      // two keys per gather (kernel-uniform)
    const u32 count = counters[cnt_idx];
    const u32 t = threadIdx.x;
    const u32 rank0 = counters[C_RANK0];
    if (t == 0) { s_cb = 0; s_ce = 0; }
    for (u64 b0 = (u64)blockIdx.x * WIN; b0 < count; b0 += (u64)gridDim.x * WIN) {
        __syncthreads();
        if (t == 0) { s_total = 0; s_fb = 0; s_fe = 0; }
        u64 rec[2] = {0, 0}, hdr[2] = {0, 0};
        u32 ax[2] = {0, 0};
        bool have[2] = {false, false};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32 e = t + k * WIN;
            if (e < TOT && b0 + e < count) {
                rec[k] = pool_rec[b0 + e]; hdr[k] = pool_hdr[b0 + e]; have[k] = true; if (!g.text) lkey[e] = (u32)(rec[k] >> (32 + KL));
                if constexpr (!W) { if (pool_aux) ax[k] = pool_aux[b0 + e]; }
            }
        }
        if (deep) {
#pragma unroll
            for (int k = 0; k < 2; ++k) if (have[k]) lkey[t + k * WIN] = (u32)rec[k];
        } else if (g.text) {       // text round: the pool records carry the suffix index only
            typename Wd<W>::sa_t fi[2] = {rec_idx<W>(rec[0]), rec_idx<W>(rec[1])};
            const bool valid[2] = {have[0] && ((hdr[0] >> 32) & 255ull) != 0, have[1] && ((hdr[1] >> 32) & 255ull) != 0};      // (len 0 = neutral entry)
            u32 key[2];
            if constexpr (!W) {      // two-stage builds: the characters in front of the suffix, for the records that come out final (GatherSpec::pc_out)
                if (g.pc_out) { ax[0] = valid[0] ? pc_fetch(g.text, (u32)fi[0]) : PC_UNKNOWN; ax[1] = valid[1] ? pc_fetch(g.text, (u32)fi[1]) : PC_UNKNOWN; }
            }
            if constexpr (!W) {
                if (dbl) {
                    u32 key2[2];
                    const u32 fi32[2] = {(u32)fi[0], (u32)fi[1]};
                    gather_keys2<2>(g, s_code, fi32, valid, key, key2);
#pragma unroll
                    for (int k = 0; k < 2; ++k) if (have[k]) lkey2[t + k * WIN] = key2[k];
                } else gather_keys<W, 2, 2>(g, s_code, fi, valid, key);
            } else gather_keys<W, 2, 2>(g, s_code, fi, valid, key);
#pragma unroll
            for (int k = 0; k < 2; ++k) if (have[k]) lkey[t + k * WIN] = key[k] >> KL;
        }
        __syncthreads();
        u32 n_lt[2], n_eq[2], n_eqb[2], lead[2];
        bool owned[2] = {false, false};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32 e = t + k * WIN;
            n_lt[k] = n_eq[k] = n_eqb[k] = 0; lead[k] = 0;
            if (have[k]) {
                const u32 sa_start = (u32)hdr[k], len = (u32)(hdr[k] >> 32) & 255u, off = (u32)(hdr[k] >> 40) & 255u;
                if (len != 0 && e >= off && e - off < WIN) {      // len 0 = neutral entry (unused chunk tail)
                    owned[k] = true;
                    const u32 ls = e - off, my = lkey[e];
                    if (deep) {
                        u32 less = 0;
                        for (u32 q = 0; q < len; ++q)
                            if (q != off && deep_other_less(g.text, g.n, my, lkey[ls + q], g.ks.depth, deep_cap, counters)) ++less;
                        sa_out[sa_start + less] = my;
                        continue;
                    }
                    bool found = false;
                    if (dbl) {                                   // ranked by (key, next key)
                        const u32 my2 = lkey2[e];
                        for (u32 q0 = 0; q0 < len; q0 += 4) {
                            u32 kq[4], kr[4];
#pragma unroll
                            for (u32 i = 0; i < 4; ++i) { const u32 a = ls + (q0 + i < len ? q0 + i : len - 1u); kq[i] = lkey[a]; kr[i] = lkey2[a]; }
#pragma unroll
                            for (u32 i = 0; i < 4; ++i) {
                                const u32 q = q0 + i;
                                if (q < len) {
                                    const bool same = kq[i] == my && kr[i] == my2;
                                    n_lt[k] += (kq[i] < my) | ((kq[i] == my) & (kr[i] < my2));
                                    if (same) { if (!found) { found = true; lead[k] = ls + q; } n_eq[k]++; n_eqb[k] += q < off; }
                                }
                            }
                        }
                    } else
                    for (u32 q0 = 0; q0 < len; q0 += 4) {       // four LDS reads in flight (the last ones clamped to the run)
                        u32 kq[4];
#pragma unroll
                        for (u32 i = 0; i < 4; ++i) kq[i] = lkey[ls + (q0 + i < len ? q0 + i : len - 1u)];
#pragma unroll
                        for (u32 i = 0; i < 4; ++i) {
                            const u32 q = q0 + i, kk = kq[i];
                            if (q < len) {
                                n_lt[k] += kk < my;
                                if (kk == my) { if (!found) { found = true; lead[k] = ls + q; } n_eq[k]++; n_eqb[k] += q < off; }
                            }
                        }
                    }
                    const u32 row = sa_start + n_lt[k] + n_eqb[k];
                    sa_out[row] = rec_idx<W>(rec[k]);
                    if constexpr (!W) { if (g.text && g.pc_out && n_eq[k] == 1u) g.pc_out[row] = ax[k]; }
                    if (mode == MODE_ISA && (n_lt[k] != 0 || ((hdr[k] >> 48) & 1ull))) isa[(u32)rec[k]] = rank0 + sa_start + n_lt[k] + 1u;
                    if (mode == MODE_DEFER) grp_out[row] = sa_start + n_lt[k];
                    if (!discard && n_eq[k] > 1 && n_eqb[k] == 0) lrun[e] = atomicAdd(&s_total, n_eq[k]);
                }
            }
        }
        __syncthreads();
        if (t == 0 && s_total) {
            const u32 need = s_total;
            if (s_cb + need > s_ce) {
                s_fb = s_cb; s_fe = s_ce;
                const u32 sz = need > chunk ? need : chunk;
                const u32 b = atomicAdd(&counters[next_cnt_idx], sz);
                if ((u64)b + sz > next_cap) { atomicOr(&counters[C_ERR], 128u); s_cb = 0; s_ce = 0; s_base = 0xffffffffu; }
                else { s_cb = b; s_ce = b + sz; }
            }
            if (s_cb + need <= s_ce) { s_base = s_cb; s_cb += need; }
        }
        __syncthreads();
        for (u32 i = s_fb + t; i < s_fe; i += 256u) { next_rec[i] = 0; next_hdr[i] = 0; }
        if (s_total == 0 || s_base == 0xffffffffu) continue;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (owned[k] && n_eq[k] > 1) {
                const u32 sa_start = (u32)hdr[k];
                const u32 o = s_base + lrun[lead[k]] + n_eqb[k];
                u64 r = (u64)rec_idx<W>(rec[k]);
                if constexpr (!W) { if (pool_aux) r = ((u64)ax[k] << 32) | (u32)r; }      // next round's key comes with the record
                next_rec[o] = r;
                next_hdr[o] = pack_hdr(sa_start + n_lt[k], n_eq[k], n_eqb[k]);
            }
    }
    __syncthreads();
    for (u32 i = s_cb + t; i < s_ce; i += 256u) { next_rec[i] = 0; next_hdr[i] = 0; }
}

// ------------------------------------------------------------------------------------------------
// Inverse suffix array for the prefix-doubling rounds (replaces the tandem-repeat machinery,
// cpp:316-484).  isa[i] = 1 + rank of suffix i; still-tied suffixes get the rank of their group head.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_isa_init(const u32* __restrict__ sa_local, const u32* __restrict__ counters,
                                                  u32* __restrict__ isa, u32 n, u32 z)
{
    const u32 ms = counters[C_MS], rank0 = counters[C_RANK0];
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < (u64)ms + z; i += (u64)gridDim.x * 256u) {
        if (i < ms) isa[sa_local[i]] = rank0 + (u32)i + 1u;
        else { const u32 j = (u32)(i - ms); isa[n - 1 - j] = j + 1u; }     // trailing 0x00 run
    }
}

// ranks of the trailing-zero suffixes (they are the z smallest, shortest first): isa[n - 1 - j] = j + 1
__global__ __launch_bounds__(256) void k_isa_zero_tail(u32* __restrict__ isa, u32 n, u32 z)
{
    for (u64 j = (u64)blockIdx.x * 256u + threadIdx.x; j < z; j += (u64)gridDim.x * 256u) isa[n - 1u - (u32)j] = (u32)j + 1u;
}

__global__ __launch_bounds__(256) void k_isa_pool(const u64* __restrict__ pool_rec, const u64* __restrict__ pool_hdr,
                                                  const u32* __restrict__ counters, u32 cnt_idx, u32* __restrict__ isa)
{
    const u32 count = counters[cnt_idx], rank0 = counters[C_RANK0];
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < count; i += (u64)gridDim.x * 256u)
        if ((pool_hdr[i] >> 32) & 255u) isa[(u32)pool_rec[i]] = rank0 + (u32)pool_hdr[i] + 1u;      // len 0 = neutral entry
}

__global__ __launch_bounds__(256) void k_isa_segs(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                  const u32* __restrict__ counters, u32* __restrict__ isa)
{
    if (blockIdx.x >= nseg) return;
    const Desc d = list[blockIdx.x];
    const u32 rank0 = counters[C_RANK0];
    const u64* src = bufs.p[d.buf & 3u] + d.rec_off;
    for (u32 p = threadIdx.x; p < d.len; p += 256u) isa[(u32)src[p]] = rank0 + d.sa_off + 1u;
}

// ------------------------------------------------------------------------------------------------
// Sharded / wide builds with deep ties: prefix doubling distributed over the shards.
//
// State of a shard between steps: its suffix-array rows and grp[r] = first row of the tie group of row r (LOCAL rows,
// relative to the slice; grp[r] == r and grp[r+1] != r for a final row).  The rank array ISA[i] = global row of the head
// of suffix i's group is REPLICATED and read-only during a step.  One step at offset h, per shard:
//   k_import_groups   rows -> tiny pool + size-class descriptor lists (identity-placed segment records)
//   k_refill / k_refill_rows   keys = ISA[i + h]
//   partition levels + LDS sorts in MODE_DEFER: rows and grp rewritten (coalesced), no rank writes
//   k_emit_updates    rows whose group head changed -> (suffix, new global head row) pairs
// then the pairs of ALL shards are exchanged (all-gatherv over RCCL, or simply applied in turn when the shards are
// logical ones on one GPU) and k_apply_updates scatters them into every replica.  Invariant kept at step boundaries:
// ISA[SA[r]] == slice_lo + grp[r] for every row of every shard.  This replaces the reference's tandem-repeat machinery
// (cpp:316-484) and its premise that buckets are independent sort problems (cpp:1652-1683) carries over: a shard only
// ever sorts its own rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grp_iota(u32* __restrict__ grp_rows, u64 nrows, u32 first_row)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < nrows; i += (u64)gridDim.x * 256u) grp_rows[i] = first_row + (u32)i;
}

__global__ __launch_bounds__(256) void k_grp_pool(const u64* __restrict__ pool_hdr, const u32* __restrict__ counters, u32 cnt_idx,
                                                  u32* __restrict__ grp_local, u32 row0)
{
    const u32 count = counters[cnt_idx];
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < count; i += (u64)gridDim.x * 256u) {
        const u64 h = pool_hdr[i];
        const u32 len = (u32)(h >> 32) & 255u, off = (u32)(h >> 40) & 255u, start = (u32)h;
        if (len) grp_local[start + off] = row0 + start;
    }
}

__global__ __launch_bounds__(256) void k_grp_segs(const Desc* __restrict__ list, u32 nseg, u32* __restrict__ grp_local, u32 row0)
{
    if (blockIdx.x >= nseg) return;
    const Desc d = list[blockIdx.x];
    for (u32 p = threadIdx.x; p < d.len; p += 256u) grp_local[d.sa_off + p] = row0 + d.sa_off;
}

// isa[suffix] = global row of its group head (row r holds rank r-1, the empty suffix sits in row 0)
template <bool W>
__global__ __launch_bounds__(256) void k_isa_from_slice(const typename Wd<W>::sa_t* __restrict__ sa_rows, const u32* __restrict__ grp, u64 rows, u64 slice_lo,
                                                        typename Wd<W>::sa_t* __restrict__ isa)
{
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x; r < rows; r += (u64)gridDim.x * 256u)
        isa[sa_rows[r]] = (typename Wd<W>::sa_t)(slice_lo + grp[r]);
}

// The same, restricted to the suffixes of one window [lo, hi) of the text: a pass per window keeps the random writes inside a
// piece of the rank array that the memory-side cache holds (narrow builds' switch to prefix doubling)
__global__ __launch_bounds__(256) void k_isa_from_slice_win(const u32* __restrict__ sa_rows, const u32* __restrict__ grp, u64 rows, u32 slice_lo,
                                                            u32* __restrict__ isa, u32 lo, u32 hi)
{
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x; r < rows; r += (u64)gridDim.x * 256u) {
        const u32 s = __builtin_nontemporal_load(sa_rows + r);
        if (s >= lo && s < hi) isa[s] = slice_lo + grp[r];
    }
}

// Rebuild tiny pool / descriptor lists from (rows, grp): the LAST row of every group emits it.  Segment records use
// identity placement (record of row r at rec[r]) and are written, with their keys, by k_refill_rows.
template <bool W>
__global__ __launch_bounds__(256) void k_import_groups(const typename Wd<W>::sa_t* __restrict__ sa_local, const u32* __restrict__ grp_local, u32 m,
                                                       const u32* __restrict__ act, u32 nact, u32 seg_buf,
                                                       u64* __restrict__ pool_rec, u64* __restrict__ pool_hdr, u32 pool_cnt_idx, u32 pool_cap,
                                                       Lists lists, Desc* __restrict__ large, u32 large_cap, u32 large_cnt_idx, u32 large_tiles_idx,
                                                       u32* __restrict__ counters)
{
    const u64 i64 = (u64)blockIdx.x * 256u + threadIdx.x;
    const bool live = i64 < (act ? nact : m);
    const u32 lp = live ? (act ? act[i64] : (u32)i64) : 0u;        // (the tail test below reads grp[lp + 1] itself: any order of `act` will do)
    u32 g = 0, len = 0;
    if (live) {
        g = grp_local[lp];
        const bool tail = (lp + 1 == m) || (grp_local[lp + 1] != g);
        len = tail ? lp - g + 1 : 0u;
    }
    {   // tiny groups: one pool allocation per wave
        const u32 want = (len >= 2 && len <= TINY_MAX) ? len : 0u;
        u32 wtot;
        const u32 woff = wave_excl_scan(want, wtot);
        u32 base = 0;
        if (wtot) {
            if (lane_id() == 0) base = atomicAdd(&counters[pool_cnt_idx], wtot);
            base = __shfl(base, 0, 64);
            if ((u64)base + wtot > pool_cap) { if (lane_id() == 0) atomicOr(&counters[C_ERR], 2u); }
            else if (want) for (u32 k = 0; k < len; ++k) { pool_rec[base + woff + k] = (u64)sa_local[g + k]; pool_hdr[base + woff + k] = pack_hdr(g, len, k); }
        }
    }
    const u32 cls = len > TINY_MAX ? class_of(len) : 4u;
    const u64 lt_mask = lane_id() ? (~0ull >> (64 - lane_id())) : 0ull;
    const Desc d = {g, len, g, DESC_BUF(32, seg_buf)};
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u64 mk = __ballot(cls == k);
        if (mk == 0) continue;
        const int leader = __ffsll((long long)mk) - 1;
        u32 base = 0;
        if ((int)lane_id() == leader) base = atomicAdd(&counters[k < 3 ? lists.cnt_idx + k : large_cnt_idx], (u32)__popcll(mk));
        base = __shfl(base, leader, 64);
        if (cls == k) {
            const u32 i = base + (u32)__popcll(mk & lt_mask);
            if (k < 3) { if (i < lists.cap[k]) lists.cls[k][i] = d; else atomicOr(&counters[C_ERR], 1u); }
            else { if (i < large_cap) large[i] = d; else atomicOr(&counters[C_ERR], 4u); }
        }
    }
    if (cls == 3) atomicAdd(&counters[large_tiles_idx], (len + P1_TILE - 1) / P1_TILE);
}

// Rank updates of one step for a window [i0, i1) of a slice's work items - rows (act == nullptr, compared with the full
// copy grp_prev[row]) or entries of the active list (compared with prev[i], k_gather_prev): every row whose group head
// differs from the one it had when the step began names a suffix whose rank changed.  An update is {suffix, new global
// head row}: one u64 (row << 32 | suffix) in narrow builds, two u64 in wide builds.
// cnt[0] = updates written (the caller sizes the window so that it cannot overflow), cnt[1] += rows still tied,
// cnt[2] = length of next step's active list act_next (tied rows, unordered), cnt[3] = 1 if it did not fit act_cap.
// (Both kernels below reserve their output room with returning atomics on ONE global counter, which the chip serves at ~90 per
// microsecond: a workgroup therefore takes UPD_K x 256 rows per reservation - with one per 256 rows the 2^33-byte config-5 stream
// spent half of its doubling time here, k_list_tied 44 ms per 268 M-row shard.)
#define UPD_K 16
template <bool W>
__global__ __launch_bounds__(256) void k_emit_updates(const typename Wd<W>::sa_t* __restrict__ sa_rows, const u32* __restrict__ grp, const u32* __restrict__ prev,
                                                      const u32* __restrict__ act, u64 rows, u64 i0, u64 i1, u64 slice_lo, u64* __restrict__ out, u64 cap,
                                                      u32* __restrict__ act_next, u64 act_cap, unsigned long long* __restrict__ cnt)
{
    __shared__ u32 s_cnt, s_tied;
    __shared__ unsigned long long s_base, s_abase;
    const u64 lt_mask = lane_id() ? (~0ull >> (64 - lane_id())) : 0ull;
    unsigned long long tied_total = 0;           // (thread 0: added to cnt[1] once, when the workgroup is done)
    for (u64 b = i0 + (u64)blockIdx.x * (256u * UPD_K); b < i1; b += (u64)gridDim.x * (256u * UPD_K)) {
        if (threadIdx.x == 0) { s_cnt = 0; s_tied = 0; }
        __syncthreads();
        u32 g[UPD_K], r[UPD_K];
        u32 mchg = 0, mtied = 0;                 // bit k: row k of this thread changed its head / is still tied
        u32 wc = 0, wt = 0;                      // what this wave writes, and where this lane's entries start inside it
        u32 oc[UPD_K], ot[UPD_K];
#pragma unroll
        for (int k = 0; k < UPD_K; ++k) {
            const u64 i = b + (u64)k * 256u + threadIdx.x;
            bool chg = false, tied = false;
            g[k] = 0; r[k] = 0;
            if (i < i1) {
                r[k] = act ? act[i] : (u32)i;
                g[k] = grp[r[k]];
                chg = g[k] != prev[i];
                tied = g[k] != r[k] || ((u64)r[k] + 1 < rows && grp[r[k] + 1] == r[k]);
            }
            const u64 mc = __ballot(chg), mt = __ballot(tied);
            oc[k] = wc + (u32)__popcll(mc & lt_mask); ot[k] = wt + (u32)__popcll(mt & lt_mask);
            wc += (u32)__popcll(mc); wt += (u32)__popcll(mt);
            mchg |= (u32)chg << k; mtied |= (u32)tied << k;
        }
        u32 wbase = 0, tbase = 0;
        if (lane_id() == 0) { if (wc) wbase = atomicAdd(&s_cnt, wc); if (wt) tbase = atomicAdd(&s_tied, wt); }
        wbase = __shfl(wbase, 0, 64); tbase = __shfl(tbase, 0, 64);
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base = s_cnt ? atomicAdd(&cnt[0], (unsigned long long)s_cnt) : 0ull;
            s_abase = 0;
            if (s_tied) {
                tied_total += s_tied;
                if (act_next) {
                    s_abase = atomicAdd(&cnt[2], (unsigned long long)s_tied);
                    if (s_abase + s_tied > act_cap) { cnt[3] = 1ull; s_abase = ~0ull; }      // next step's list overflows: ONE store says so, nothing of this tile is listed
                }
            }
        }
        __syncthreads();
        const bool list_next = act_next != nullptr && s_abase != ~0ull;          // (workgroup-uniform)
#pragma unroll
        for (int k = 0; k < UPD_K; ++k) {
            if ((mchg >> k) & 1u) {
                const u64 o = s_base + wbase + oc[k];
                if (o < cap) {
                    if constexpr (W) { out[2 * o] = sa_rows[r[k]]; out[2 * o + 1] = slice_lo + g[k]; }
                    else out[o] = ((slice_lo + g[k]) << 32) | (u64)sa_rows[r[k]];
                }
            }
            if (((mtied >> k) & 1u) && list_next) act_next[s_abase + tbase + ot[k]] = r[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && tied_total) atomicAdd(&cnt[1], tied_total);
}

// list of the tied rows of a slice (any order): lets the very first doubling step run on the list instead of on all rows.
// cnt[2] = entries written, cnt[3] = 1 if they did not fit
__global__ __launch_bounds__(256) void k_list_tied(const u32* __restrict__ grp, u64 rows, u32* __restrict__ act, u64 act_cap, unsigned long long* __restrict__ cnt)
{
    __shared__ u32 s_tied;
    __shared__ unsigned long long s_base;
    const u64 lt_mask = lane_id() ? (~0ull >> (64 - lane_id())) : 0ull;
    for (u64 b = (u64)blockIdx.x * (256u * UPD_K); b < rows; b += (u64)gridDim.x * (256u * UPD_K)) {
        if (threadIdx.x == 0) s_tied = 0;
        __syncthreads();
        u32 mtied = 0, wt = 0, ot[UPD_K];
#pragma unroll
        for (int k = 0; k < UPD_K; ++k) {
            const u64 r = b + (u64)k * 256u + threadIdx.x;
            bool tied = false;
            if (r < rows) { const u32 g = grp[r]; tied = g != (u32)r || (r + 1 < rows && grp[r + 1] == (u32)r); }
            const u64 mt = __ballot(tied);
            ot[k] = wt + (u32)__popcll(mt & lt_mask);
            wt += (u32)__popcll(mt);
            mtied |= (u32)tied << k;
        }
        u32 tbase = 0;
        if (lane_id() == 0 && wt) tbase = atomicAdd(&s_tied, wt);
        tbase = __shfl(tbase, 0, 64);
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base = s_tied ? atomicAdd(&cnt[2], (unsigned long long)s_tied) : 0ull;
            // the list does not hold this tile: ONE store says so (round 6: every lane of every overflowing tile stored the flag - hundreds
            // of millions of stores to one address on a slice that is mostly tied: 4.9 ms per 2^28-row shard of the 8 GiB DNA job)
            if (s_tied && s_base + s_tied > act_cap) cnt[3] = 1ull;
        }
        __syncthreads();
        if (s_tied && s_base + s_tied > act_cap) break;          // (workgroup-uniform; the counter only grows: every later tile overflows as well,
                                                                 // and a list that overflowed is not used)
#pragma unroll
        for (int k = 0; k < UPD_K; ++k)
            if ((mtied >> k) & 1u) act[s_base + tbase + ot[k]] = (u32)(b + (u64)k * 256u + threadIdx.x);
        __syncthreads();
    }
}

// prev[i] = group head of active row act[i] when the step begins
__global__ __launch_bounds__(256) void k_gather_prev(const u32* __restrict__ grp, const u32* __restrict__ act, u64 nact, u32* __restrict__ prev)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < nact; i += (u64)gridDim.x * 256u) prev[i] = grp[act[i]];
}

template <bool W>
__global__ __launch_bounds__(256) void k_apply_updates(const u64* __restrict__ upd, u64 count, typename Wd<W>::sa_t* __restrict__ isa)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < count; i += (u64)gridDim.x * 256u) {
        if constexpr (W) isa[upd[2 * i]] = upd[2 * i + 1];
        else { const u64 u = upd[i]; isa[(u32)u] = (u32)(u >> 32); }
    }
}

// SA[0] = n and the trailing-zero-run rows (descending index)
// ------------------------------------------------------------------------------------------------
// Tandem repeats (the reference's partition_tandem_repeats / complete_tandem_repeats, cpp:316-484, in the shape that suits
// prefix doubling).  At doubling offset h a tie group = all suffixes that share their first h characters.  If two members
// sit P <= h positions apart, the text has period P from the first of them to h characters behind the second.  If the members
// of a group are ONE arithmetic progression i0, i0 + P, ..., last (a single run of a repeated unit: the usual case for units
// of ten characters and more), the text repeats with period P from i0 to last + h, so suffix(i) and suffix(i + P) agree until
// the repeat ends and first differ at the same text position for EVERY member - the members are ordered by position, all
// one way, and the way is the order of suffix(last + P) against suffix(last): those two already differ inside their first h
// characters (last + P is not in the group), so the rank array knows it.  No character is read.
// Doubling alone needs log2(run length / h) more rounds over the whole group (DNA with tandem repeats: 78 % of all suffixes
// stay tied for eight rounds); this kernel finishes such a group at once: final rows (and ranks, or group heads) are written
// and the descriptor is neutralised (len = 0).
// One workgroup per segment (persistent).  Membership tests go through the rank array: same group <=> same rank value.
//   MODE_ISA   (narrow, in place): members of a segment emitted by the previous round hold 1 + the group's first row;
//              ranks are updated here, the finished records are marked for k_refill.
//   MODE_DEFER (sharded / wide): the rank array is read-only during a step; grp_out[row] = row marks the rows final and the
//              rank updates are derived from that afterwards (k_emit_updates).
// ------------------------------------------------------------------------------------------------
template <int THREADS, int ITEMS, bool W>
__global__ __launch_bounds__(THREADS) void k_chain_resolve(RecBufs bufs, Desc* __restrict__ list, u32 nseg, typename Wd<W>::sa_t* __restrict__ sa_rows,
                                                           typename Wd<W>::sa_t* __restrict__ isa_rw, const typename Wd<W>::sa_t* __restrict__ isa_ro,
                                                           u32* __restrict__ grp_out, u32 mode, u64 n, u64 h, u32* __restrict__ counters)
{
    typedef typename Wd<W>::sa_t idx_t;
    constexpr idx_t NONE = ~(idx_t)0;
    const idx_t* isa = mode == MODE_ISA ? isa_rw : isa_ro;
    __shared__ u32 s_P, s_nf;
    __shared__ unsigned long long s_tail;
    const u32 t = threadIdx.x;
    const u32 rank0 = counters[C_RANK0];
    const u32 wmax = h < 64u ? (u32)h : 64u;                 // steps looked for
    for (u32 s = blockIdx.x; s < nseg; s += gridDim.x) {
        const Desc d = list[s];
        __syncthreads();
        if (d.len < 4u || d.len > (u32)(THREADS * ITEMS) || (d.buf & DESC_STALE)) continue;      // (workgroup-uniform)
        if (t == 0) { s_P = 0xffffffffu; s_nf = 0; s_tail = ~0ull; }
        __syncthreads();
        // (the step is looked for among the first rows: the others are loaded once there is one - most segments of the later doubling
        // rounds have none, and a class-B segment is up to 18 KB of rows)
        idx_t idx[ITEMS];
#pragma unroll
        for (int j = 1; j < ITEMS; ++j) idx[j] = NONE;
        idx[0] = t < d.len ? sa_rows[d.sa_off + t] : NONE;
        const idx_t i00 = sa_rows[d.sa_off];
        if ((u64)i00 >= n) continue;
        const idx_t g0 = isa[i00];                            // the rank every member holds
        // Membership test used below: "suffix j is in this group <=> isa[j] == g0".  It rests on two things.  (a) No stale
        // sibling: DESC_STALE segments (fresh from a partition level of this round, members still holding the PARENT's rank)
        // were skipped above, so g0 names this group alone.  (b) In MODE_ISA other workgroups rewrite isa[] for THEIR groups
        // while this one reads it (they finish their own progressions): the values they write are rows inside their own
        // groups' row ranges, and the row ranges of different groups are disjoint - no rewrite can produce or destroy the
        // value g0, so a racing read answers the membership question as if it had come before or after.  The direction
        // test (step 3) reads the rank of a suffix OUTSIDE the group: a rewrite moves it inside its own group's row range,
        // on the same side of g0.  (MODE_DEFER: the rank array is read-only during a step.)
        // 1. the step: the first 64 members look for their next member within wmax positions (units longer than 64 - and groups
        // that are several progressions, tiny or class-L groups - are left to the doubling rounds: correct, log2 more rounds)
        if (t < 64u && t < d.len) {
            const u64 i = idx[0];
            u32 found = 0xffffffffu;
            if (i < n && isa[i] == g0) {
                for (u32 q0 = 1; q0 <= wmax && found == 0xffffffffu; q0 += 8) {
                    idx_t v[8];
#pragma unroll
                    for (u32 k = 0; k < 8; ++k) v[k] = (q0 + k <= wmax && i + q0 + k < n) ? isa[i + q0 + k] : (idx_t)0;
#pragma unroll
                    for (u32 k = 0; k < 8; ++k) if (found == 0xffffffffu && v[k] == g0) found = q0 + k;
                }
            } else found = 0u;                                // (not the ranks this kernel expects: leave the segment alone)
            if (found != 0xffffffffu) atomicMin(&s_P, found);
        }
        __syncthreads();
        const u32 P = s_P;
        if (P == 0u || P == 0xffffffffu) continue;
#pragma unroll
        for (int j = 1; j < ITEMS; ++j) { const u32 p = (u32)j * THREADS + t; idx[j] = p < d.len ? sa_rows[d.sa_off + p] : NONE; }
        // 2. one progression?  every member but one must have its successor i + P in the group
        u32 nf = 0;
        u64 tail = ~0ull;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (idx[j] != NONE) {
                const bool f = (u64)idx[j] + P < n && isa[(u64)idx[j] + P] == g0;
                nf += f;
                if (!f) tail = (u64)idx[j] < n ? (u64)idx[j] : 0ull;      // (0: no progression can end there)
            }
        nf = wave_sum(nf);
        if ((t & 63u) == 0 && nf) atomicAdd(&s_nf, nf);
        if (tail != ~0ull) atomicMin(&s_tail, (unsigned long long)tail);      // (more than one: rejected by the count)
        __syncthreads();
        if (s_nf != d.len - 1u) continue;
        const u64 last = s_tail;
        if (last < (u64)(d.len - 1u) * P) continue;
        const u64 first = last - (u64)(d.len - 1u) * P;
        // 3. the direction: suffix(last + P) against suffix(last), as the ranks have it (rank 0 behind the end of the text)
        const idx_t rx = last + P < n ? isa[last + P] : (idx_t)0;
        const bool descending = rx < g0;                      // later positions are the smaller suffixes
        // 4. final rows
        bool okp = true;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (idx[j] != NONE) {
                const u64 off = (u64)idx[j] - first;
                const u32 k = (u32)(off / P);
                okp &= (u64)idx[j] >= first && (u64)k * P == off && k < d.len;
            }
        if (!__syncthreads_and(okp)) continue;                // (not the progression it seemed to be: doubling goes on)
        u64* src = bufs.p[d.buf & 3u] + d.rec_off;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (idx[j] != NONE) {
                const u32 k = (u32)(((u64)idx[j] - first) / P);
                const u32 r = descending ? d.len - 1u - k : k;
                sa_rows[d.sa_off + r] = idx[j];
                if (mode == MODE_ISA) {
                    isa_rw[idx[j]] = (idx_t)(rank0 + d.sa_off + r + 1u);
                    src[(u32)j * THREADS + t] = ~0ull;        // a finished record: k_refill, which sweeps the whole record array, skips it
                } else grp_out[d.sa_off + r] = d.sa_off + r;
            }
        if (t == 0) { list[s].len = 0; atomicAdd(&counters[C_CHAIN], d.len); }
    }
}

template <bool W>
__global__ __launch_bounds__(256) void k_sa_head(typename Wd<W>::sa_t* __restrict__ sa, u64 n, u64 z)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < (z ? z : 1); i += (u64)gridDim.x * 256u) {
        if (i == 0) sa[0] = (typename Wd<W>::sa_t)n;
        if (i < z) sa[1 + i] = (typename Wd<W>::sa_t)(n - 1 - i);
    }
}
