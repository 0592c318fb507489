// Two-stage suffix-array build for text-like inputs: sort only the B* suffixes with the
// engine of sa_kernels.hip.h, then INDUCE the rest.  Counterparts in the reference: suffix typing + B* scatter
// msufsort.cpp:1496-1555, second stage (right-to-left pass for the B suffixes, left-to-right pass for the A suffixes)
// cpp:646-791 and cpp:867-1017, driven from cpp:1021-1057.  The reference walks each pass with a few threads that hand
// batches to each other; here a pass is a sequence of data-parallel STABLE SCATTERS:
//
//   suffix i is type B if T[i..] < T[i+1..], else type A (the suffix of the last character is A); B* = B followed by A.
//   Row layout of first-byte bucket c0:  [ A suffixes | B suffixes by second byte c1 >= c0: (c0,c0) (c0,c0+1) ... ],
//   inside sub-bucket (c0,c1): [ B* | other B ].
//
//   pass B, c1 = 255 .. 0: the B rows of bucket c1, read right to left, put every B-type predecessor j-1 (T[j-1] <= c1) at
//       the right end of sub-bucket (T[j-1], c1).  All sources of one bucket except those that land in (c1,c1) itself are
//       known when the bucket starts, so "level 0" = sub-buckets (c1, > c1) is ONE stable scatter by T[j-1]; what it puts
//       into (c1,c1) is level 1, and so on: as many levels as the longest run of the byte c1 in the text.
//   pass A, c0 = 0 .. 255: the rows of bucket c0, read left to right, put every A-type predecessor at the left end of the A
//       area of bucket T[j-1] >= c0: levels over the A area (what lands in A(c0) itself feeds the next level), then the B area.
//
// Every row carries characters in front of its suffix (pc[row] = T[j-1] | older << 8 | count << 24, j = SA[row]), so a source costs
// no text access to find its target, and the row it induces inherits the remaining characters.  `older` (16 bits): T[j-2], T[j-3] as
// they are - or, for alphabets of up to 16 byte values (IndTables::pc_bits < 8, round 6), the DENSE NUMBERS of T[j-2] .. T[j-1-K],
// K = 16 / pc_bits: 4 older characters for 16 values, 8 for a DNA.  The text is read (one unaligned load) only for B* rows the
// first stage left without characters and for every (K + 1)-th row of an induction chain - every third row with plain bytes, every
// ninth for a DNA (a fetch is a random 64-byte sector: 17 - 22 ps each at the rate the levels sustain,
// profiles/r06_induction_lookback_window_and_fetch_slope.txt).
// Random text accesses of the whole second stage: about 1.3 per B* suffix, each a 128-byte line of HBM traffic (against a line
// per tied suffix and key round in the sort-all path).
#pragma once
#include "sa_kernels.hip.h"

#define TY_THREADS 256
#define TY_TILE (TY_THREADS * 32)          // positions per workgroup: every thread owns one 32-bit word of the bitmaps
#define TY_SCAN_LIMIT (1u << 16)           // give up on runs of one byte longer than this (flag: the caller sorts all suffixes)
#define IND_MAXRUN_CAP 4096u

#define IND_FLAG_LONGRUN 1u
#define IND_FLAG_CURSOR 2u           // pass B
#define IND_FLAG_CURSOR_A 4u

// ---- suffix types: bitmap of the B* positions ----
// diag[c] += B positions whose next byte is the same byte c.  (With different bytes the pair decides the type: the 16-bit
// histogram of all suffixes already is the histogram of the B suffixes above the diagonal, and zero below it.)
// Since round 6 also what k_maxrun counted in a pass of its own: the longest run of every byte value (2 or more; = the number of levels
// an induction pass needs inside that byte's bucket) and runs[c][k] = runs of byte c that are k + 2 long (k = 7: nine or more), the
// bounds for the sizes of the deeper levels.  A run belongs to the thread that holds its FIRST byte; the type bits F already say
// where neighbouring bytes differ.
__global__ __launch_bounds__(TY_THREADS) void k_types(const u8* __restrict__ text, u64 n, u32* __restrict__ bs_bits,
                                                      u32* __restrict__ flags, u32* __restrict__ diag /* 256, zeroed */,
                                                      u32* __restrict__ maxrun /* 256, zeroed */, u32* __restrict__ runs /* 256 x 8, zeroed */)
{
    __shared__ u32 s_has[TY_THREADS / 64], s_first[TY_THREADS / 64];
    __shared__ u32 s_carry;
    __shared__ u32 s_diag[256];
    __shared__ u32 s_max[256];
    __shared__ u32 s_runs[256 * 8];
    const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    s_diag[t] = 0;
    s_max[t] = 0;
    for (u32 i = t; i < 2048u; i += TY_THREADS) s_runs[i] = 0;
    // (a workgroup takes many tiles and adds its diagonal counts to the global ones once: every such add is an atomic on one
    // of a few dozen addresses, which the memory system serves one after the other)
    const u64 ntiles = (n + TY_TILE - 1) / TY_TILE;
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    const u64 tile0 = tile * TY_TILE;
    const u64 base = tile0 + (u64)t * 32u;
    u32 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = 0;
    if (base < n) {                                   // (the text is padded with 64 zero bytes)
        const uint4 a = *reinterpret_cast<const uint4*>(text + base), b = *reinterpret_cast<const uint4*>(text + base + 16);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        w[8] = *reinterpret_cast<const u32*>(text + base + 32);
    }
    // F: the type is decided at this position (next byte differs, or last position); L: ... and it is B
    u32 F = 0, L = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const u32 a = (w[i >> 2] >> (8 * (i & 3))) & 255u, b = (w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 255u;
        const u64 pos = base + i;
        const bool tail = pos + 1 >= n;               // the last suffix is type A; positions behind the text read as A
        F |= (u32)(tail || a != b) << i;
        L |= (u32)(!tail && a < b) << i;
    }
    const bool has = F != 0;
    const u32 first = has ? (L >> (__ffs((int)F) - 1)) & 1u : 0u;
    const u64 has_m = __ballot(has), first_m = __ballot(first != 0);
    if (lane == 0) { s_has[wv] = has_m != 0; s_first[wv] = has_m ? (u32)((first_m >> (__ffsll((long long)has_m) - 1)) & 1ull) : 0u; }
    __syncthreads();
    if (wv == 0) {      // type of the first position behind this tile (decides the last position's B* bit, and the types of a run
                        // of equal bytes that crosses the boundary)
        u32 carry = 0;
        {
            u64 e = tile0 + TY_TILE;
            const u64 e0 = e;
            for (;;) {
                const u64 pos = e + lane;
                const u32 a = pos < n ? text[pos] : 0u, b = pos + 1 < n ? text[pos + 1] : 0u;
                const bool tail = pos + 1 >= n;
                const u64 f = __ballot(tail || a != b), l = __ballot(!tail && a < b);
                if (f) { carry = (u32)((l >> (__ffsll((long long)f) - 1)) & 1ull); break; }
                e += 64;
                if (e - e0 > TY_SCAN_LIMIT) { if (lane == 0) atomicOr(flags, IND_FLAG_LONGRUN); break; }
            }
        }
        if (lane == 0) s_carry = carry;
    }
    __syncthreads();
    // type of the position behind my 32
    u32 carry;
    const u64 above = lane == 63 ? 0ull : (has_m & (~0ull << (lane + 1)));
    if (above) carry = (u32)((first_m >> (__ffsll((long long)above) - 1)) & 1ull);
    else {
        carry = s_carry;
        for (int w2 = TY_THREADS / 64 - 1; w2 > (int)wv; --w2) if (s_has[w2]) carry = s_first[w2];
    }
    u32 tb = 0, cur = carry;
#pragma unroll
    for (int i = 31; i >= 0; --i) {
        cur = ((F >> i) & 1u) ? ((L >> i) & 1u) : cur;
        tb |= cur << i;
    }
    const u32 nxt = (tb >> 1) | (carry << 31);
    bs_bits[base >> 5] = tb & ~nxt;
    // byte i of my 32 (i varies per lane: the word is picked with selects, indexing w[] with it would put the array into scratch)
    auto byte_at = [&](u32 i) {
        const u32 q = i >> 2;
        const u32 lo4 = q & 2u ? (q & 1u ? w[3] : w[2]) : (q & 1u ? w[1] : w[0]), hi4 = q & 2u ? (q & 1u ? w[7] : w[6]) : (q & 1u ? w[5] : w[4]);
        return ((q & 4u ? hi4 : lo4) >> (8u * (i & 3u))) & 255u;
    };
    const u32 eq = tb & ~F;                           // B positions inside a run of equal bytes
    if (__ballot(eq != 0)) {                          // (static indexing; a loop over the runs of set bits with byte_at: DNA 1.07 -> 1.17 ms per GiB)
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if ((eq >> i) & 1u) atomicAdd(&s_diag[(w[i >> 2] >> (8 * (i & 3))) & 255u], 1u);
    }
    // runs of two or more equal bytes that START in my 32 positions: R bit i = byte i equals byte i + 1 (both inside the text)
    {
        const u32 R = ~F;
        u32 before = (u32)__shfl_up((int)(w[7] >> 24), 1, 64);               // the byte in front of my first one
        if (lane == 0) before = base != 0 && base < n ? text[base - 1] : 0x100u;
        const u32 cont0 = before == (w[0] & 255u) ? 1u : 0u;
        u32 S2 = R & ~((R << 1) | cont0);
        {
            while (S2) {                             // (a text: 0.6 starts per thread)
                    const u32 i = (u32)__ffs((int)S2) - 1u;
                    S2 &= S2 - 1u;
                    const u32 ch = byte_at(i);
                    const u32 rest = ~(R >> i);
                    const u32 ones = rest ? (u32)__ffs((int)rest) - 1u : 32u;      // equal neighbours from i on, inside my word
                    u32 len = ones + 1u;
                    if ((u32)i + ones == 32u) {      // the run reaches my last byte and the first one behind it: how far does it go?
                        bool open = true;
#pragma unroll
                        for (int k = 1; k < 4; ++k)
                            if (open) { if (base + 32u + (u32)k < n && ((w[8] >> (8 * k)) & 255u) == ch) ++len; else open = false; }
                        if (open) {
                            u64 p = base + 36u;
                            while (p < n && text[p] == ch && len < IND_MAXRUN_CAP) { ++p; ++len; }
                        }
                    }
                    if (len >= IND_MAXRUN_CAP) atomicOr(flags, IND_FLAG_LONGRUN);
                    if (len > s_max[ch]) atomicMax(&s_max[ch], len);
                    atomicAdd(&s_runs[ch * 8u + (len > 9u ? 9u : len) - 2u], 1u);
                }
        }
    }
    }
    __syncthreads();
    if (s_diag[t]) atomicAdd(&diag[t], s_diag[t]);
    if (s_max[t]) atomicMax(&maxrun[t], s_max[t]);
    for (u32 i = t; i < 2048u; i += TY_THREADS) if (s_runs[i]) atomicAdd(&runs[i], s_runs[i]);
}

// ---- tables of the row layout (made on the host from the three histograms) ----
struct IndTables {
    const u32* bkt;         // [257] first row of every first-byte bucket (row 0 = the empty suffix)
    const u32* aend;        // [256] first row behind the A area = first row of the B area
    const u32* sub_start;   // [65536] first row of sub-bucket (c0,c1); B* first
    const u32* sub_cnt;     // [65536] B suffixes of (c0,c1)
    const u32* sub_bs;      // [65536] B* suffixes of (c0,c1)
    const u32* bs_off;      // [65536] where the B* suffixes of (c0,c1) start in the sorted B* array
    const u8* code;         // [256] dense number of every byte value that occurs in the text (others: 255)
    const u8* sym;          // [nb] the byte value of every dense number
    u32 nb;                 // dense numbers, rounded up to a multiple of 8: per-tile counts are kept as [tile][nb]
    u32 nsym;               // byte values in use
    u32 pc_bits;            // bits per older character in pc[] (2 .. 5: dense numbers, 5 by switch only; 8: plain bytes)
};

#define IND_MAX_LEVELS 32768u
struct IndState {
    u32 cur[256];           // pass B: next free row + 1 (right end, exclusive) of sub-bucket (c0, current c1); pass A: next free row of A(c)
    u32 rng[2][2];          // source rows [lo, hi) of the level being processed / of the next level
    u32 flags;
    u32 spin_limit;         // bound of the look-back spins (host: 2^22; MSUFSORT_HIP_IND_SPIN shrinks it for the time-out test)
    u32 decided[2];         // level launches that settled on fixed hand-out / on the ticket counter (statistics)
    u32 ticket[IND_MAX_LEVELS];   // single-pass levels: next tile to hand out, one counter per level launch (zeroed with the state)
    u32 mode[IND_MAX_LEVELS];     // ... and how the tiles behind the first round are handed out: 0 undecided, IND_MODE_FIXED, IND_MODE_TICKET
};
#define IND_MODE_FIXED 1u
#define IND_MODE_TICKET 2u

#define IND_FLAG_LOOKBACK 8u      // a look-back waited longer than any kernel runs: the build is reported as failed, not hung
// tile status of the single-pass levels, one 64-bit word per (tile, byte value in use): [63:62] 1 = this tile's count,
// 2 = count of all tiles up to and including this one; [61:48] level launch number (stale words of earlier levels read as
// "nothing yet"); [47] POISON: a tile in front of this one lost its look-back (its running total is short), nothing computed
// from it may be written - the tiles behind learn it from the words they read anyway, not from another load; [46:0] the count
#define IND_ST_POISON (1ull << 47)
#define IND_ST_AGG 1ull
#define IND_ST_INC 2ull
__device__ __forceinline__ bool ind_failed(const IndState* st) { return (__hip_atomic_load(&st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & IND_FLAG_LOOKBACK) != 0; }
__device__ __forceinline__ u64 ind_status(u64 flag, u32 epoch, u64 value) { return (flag << 62) | ((u64)(epoch & 0x3fffu) << 48) | value; }

// sorted B* suffixes -> the left ends of their sub-buckets; one workgroup per non-empty (c0,c1)
// ... and what the first stage already knows about the characters in front of them (spc: GatherSpec::pc_out of the sorts, PC_UNKNOWN where
// a suffix became final without a gather; nullptr: nothing is known) - the first level of pass B fetches only the unknown ones
__global__ __launch_bounds__(256) void k_place_bstar(const u32* __restrict__ sstar, const u32* __restrict__ spc, const u32* __restrict__ keys, IndTables tb,
                                                     u32* __restrict__ sa, u32* __restrict__ pc)
{
    const u32 key = keys[blockIdx.x];
    const u32 cnt = tb.sub_bs[key], src = tb.bs_off[key], dst = tb.sub_start[key];
    const u32 bits = tb.pc_bits;
    for (u32 i = blockIdx.y * 256u + threadIdx.x; i < cnt; i += gridDim.y * 256u) {
        sa[dst + i] = sstar[src + i];
        u32 v = spc ? spc[src + i] : PC_UNKNOWN;
        if (v != PC_UNKNOWN && bits != 8u) {         // the sorts leave plain bytes (pc_fetch): the older ones become dense numbers
            const u32 n_ch = v >> 24;
            u32 older = 0;
            if (n_ch >= 2u) older |= (u32)tb.code[(v >> 8) & 255u];
            if (n_ch >= 3u) older |= (u32)tb.code[(v >> 16) & 255u] << bits;
            v = (v & 255u) | (older << 8) | (n_ch << 24);
        }
        pc[dst + i] = v;
    }
}

// the (up to) three characters in front of suffix j and how many there are: pc_fetch (sa_kernels.hip.h; the sorts of the first stage
// read them next to the keys they gather)
__device__ __forceinline__ u32 ind_fetch(const u8* __restrict__ text, u32 j) { return pc_fetch(text, j); }
// ... only T[j-1]: valid whatever the format of the older characters (count 1: the row it induces fetches its own)
__device__ __forceinline__ u32 ind_fetch1(const u8* __restrict__ text, u32 j) { return j ? ((u32)text[j - 1u] | (1u << 24)) : 0u; }
// ... in the format of the level: T[j-1], then up to K = 16 / bits older characters as dense numbers (codes: byte -> number, LDS)
__device__ __forceinline__ u32 ind_fetch_p(const u8* __restrict__ text, u32 j, u32 bits, const u8* codes)
{
    if (bits == 8u) return pc_fetch(text, j);
    const u32 K = 16u / bits;
    if (j < 12u) {                                   // (the first bytes of the text: no 12-byte window in front)
        if (j == 0u) return 0u;
        const u32 cnt = j < K + 1u ? j : K + 1u;
        u32 older = 0;
        for (u32 k = 1; k < cnt; ++k) older |= (u32)codes[text[j - 1u - k]] << (bits * (k - 1u));
        return (u32)text[j - 1u] | (older << 8) | (cnt << 24);
    }
    u32 w[3];
    __builtin_memcpy(w, text + j - 12u, 12);         // byte 11 = T[j-1], byte 11 - k = T[j-1-k]
    u32 older = 0;
#pragma unroll
    for (int k = 1; k <= 8; ++k)
        if ((u32)k <= K) older |= (u32)codes[(w[(11 - k) >> 2] >> (8 * ((11 - k) & 3))) & 255u] << (bits * (u32)(k - 1));
    return (w[2] >> 24) | (older << 8) | ((K + 1u) << 24);
}

#define IND_ITEMS 16
#define IND_TILE (256u * IND_ITEMS)      // sources per tile: 256 threads x 16, wave-major rows of 64

// start of a bucket's work
//   kind 0 (pass B, bucket c): cursors = right ends of the sub-buckets (c0, c); level 0 = sub-buckets (c, > c)
//   kind 1 (pass A, bucket c, A levels): level 0 = what earlier buckets put into A(c)
//   kind 2 (pass A, bucket c, B area)
//   kind 3 (start of pass A): cursors = left ends of the A areas; the empty suffix (row 0) induces suffix n-1
__global__ __launch_bounds__(256) void k_ind_setup(IndState* __restrict__ st, IndTables tb, u32 kind, u32 c, const u8* __restrict__ text, u32 n,
                                                   u32* __restrict__ sa, u32* __restrict__ pc)
{
    const u32 t = threadIdx.x;
    if (kind == 0) {
        if (t <= c) st->cur[t] = tb.sub_start[t * 256u + c] + tb.sub_cnt[t * 256u + c];
        if (t == 0) { st->rng[0][0] = tb.sub_start[c * 256u + c] + tb.sub_cnt[c * 256u + c]; st->rng[0][1] = tb.bkt[c + 1]; }
    } else if (kind == 1) {
        if (t == 0) { st->rng[0][0] = tb.bkt[c]; st->rng[0][1] = st->cur[c]; }
    } else if (kind == 2) {
        if (t == 0) { st->rng[0][0] = tb.aend[c]; st->rng[0][1] = tb.bkt[c + 1]; }
    } else {
        u32 v = tb.bkt[t];
        const u32 last = text[n - 1];
        if (t == last) { sa[v] = n - 1; pc[v] = ind_fetch1(text, n - 1); ++v; }
        st->cur[t] = v;
        if (t == 0) sa[0] = n;
    }
}

struct IndLevel {
    u32 pass_b;      // 1: right-to-left pass over B rows; 0: left-to-right pass
    u32 c;           // bucket being read
    u32 slot;        // which rng[] holds this level's rows
    u32 stars;       // pass B level 0: the sources include B* rows, whose pc is not known yet
    u32 src_a;       // pass A: the sources are A rows (an equal preceding byte is then type A too)
    u32 fixed_tiles; // (unused since round 4: every launch settles fixed / ticket hand-out itself, see k_ind_fused)
};

// target bin of one source row, or 256: nothing to induce
__device__ __forceinline__ u32 ind_bin(const IndLevel& lv, u32 j, u32 pcw)
{
    if (j == 0) return 256u;
    const u32 pcv = pcw & 255u;
    if (lv.pass_b) return pcv <= lv.c ? pcv : 256u;
    return (pcv > lv.c || (pcv == lv.c && lv.src_a)) ? pcv : 256u;
}

// per tile and target bin: number of rows this level writes (also fetches the preceding character of B* sources)
__global__ __launch_bounds__(256) void k_ind_count(const IndState* __restrict__ st, IndLevel lv, IndTables tb, const u32* __restrict__ sa, u32* __restrict__ pc,
                                                   const u8* __restrict__ text, u32* __restrict__ tile_hist)
{
    __shared__ u32 hist[8 * 256];
    __shared__ u32 s_sub[257];           // pass B level 0: first rows of the sub-buckets of bucket c (binary search: which one holds a row)
    const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const u32 lo = st->rng[lv.slot][0], hi = st->rng[lv.slot][1];
    const u32 cnt = hi - lo, ntiles = (cnt + IND_TILE - 1) / IND_TILE;
    if (blockIdx.x >= ntiles) return;
    if (lv.stars) { s_sub[t] = tb.sub_start[lv.c * 256u + t]; if (t == 0) s_sub[256] = tb.bkt[lv.c + 1]; }
    const u32 my_code = tb.code[t];
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (u32 q = 0; q < 8; ++q) hist[q * 256u + t] = 0;
        __syncthreads();
        u32 j[IND_ITEMS], pcv[IND_ITEMS], row[IND_ITEMS];
        bool star[IND_ITEMS];
#pragma unroll
        for (int i = 0; i < IND_ITEMS; ++i) {
            const u32 q = tile * IND_TILE + wv * (IND_ITEMS * 64u) + i * 64u + lane;
            row[i] = 0xffffffffu; j[i] = 0; pcv[i] = 0; star[i] = false;
            if (q < cnt) {
                const u32 r = lv.pass_b ? hi - 1u - q : lo + q;
                row[i] = r;
                j[i] = sa[r];
                if (lv.stars) {
                    u32 a = lv.c, b = 256u;              // sub-bucket c2 with s_sub[c2] <= r < s_sub[c2 + 1]
                    while (b - a > 1u) { const u32 mid = (a + b) >> 1; if (s_sub[mid] <= r) a = mid; else b = mid; }
                    star[i] = r < s_sub[a] + tb.sub_bs[lv.c * 256u + a];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < IND_ITEMS; ++i)
            if (row[i] != 0xffffffffu) { pcv[i] = pc[row[i]]; star[i] = star[i] && pcv[i] == PC_UNKNOWN; }
#pragma unroll
        for (int i = 0; i < IND_ITEMS; ++i)
            if (row[i] != 0xffffffffu && star[i]) pcv[i] = ind_fetch1(text, j[i]);      // (diagnostic path: one character, any format)
#pragma unroll
        for (int i = 0; i < IND_ITEMS; ++i) {
            const bool in = row[i] != 0xffffffffu;
            if (in && star[i]) pc[row[i]] = pcv[i];
            const u32 b = in ? ind_bin(lv, j[i], pcv[i]) : 256u;
            // (eight copies of the bins, copy = lane & 7: text and DNA put everything on a handful of bytes, and lanes that
            // meet on one LDS address are served one after the other)
            if (b < 256u) atomicAdd(&hist[b * 8u + (lane & 7u)], 1u);          // (copies of one byte value next to each other: eight banks)
        }
        __syncthreads();
        if (my_code != 255u || tb.nb == 256u) {      // (255 = byte value that does not occur, unless all do)
            u32 sum = 0;
#pragma unroll
            for (u32 q = 0; q < 8; ++q) sum += hist[t * 8u + q];
            tile_hist[(u64)tile * tb.nb + my_code] = sum;
        }
    }
}

// per target byte (one workgroup per dense number): exclusive prefix over the tiles -> absolute first target row of every tile;
// moves the cursor; the rows that land in the bucket being read are the next level
#define IND_SCAN_PER 8
__global__ __launch_bounds__(256) void k_ind_scan(IndState* __restrict__ st, IndLevel lv, IndTables tb, u32* __restrict__ tile_hist)
{
    __shared__ u32 wsum[4];
    __shared__ u32 s_run;
    const u32 code = blockIdx.x, bin = tb.sym[code], nb = tb.nb;
    const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const u32 lo = st->rng[lv.slot][0], hi = st->rng[lv.slot][1];
    const u32 ntiles = (hi - lo + IND_TILE - 1) / IND_TILE;
    const u32 base = st->cur[bin];
    if (t == 0) s_run = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < ntiles; t0 += 256u * IND_SCAN_PER) {
        const u32 first = t0 + t * IND_SCAN_PER;
        u32 v[IND_SCAN_PER], sum = 0;
#pragma unroll
        for (int k = 0; k < IND_SCAN_PER; ++k) { v[k] = first + k < ntiles ? tile_hist[(u64)(first + k) * nb + code] : 0u; sum += v[k]; }
        u32 wt;
        const u32 e = wave_excl_scan(sum, wt);
        if (lane == 63) wsum[wv] = wt;
        __syncthreads();
        u32 before = s_run + e;
        for (u32 w2 = 0; w2 < wv; ++w2) before += wsum[w2];
#pragma unroll
        for (int k = 0; k < IND_SCAN_PER; ++k) {
            if (first + k < ntiles) tile_hist[(u64)(first + k) * nb + code] = lv.pass_b ? base - 1u - before : base + before;
            before += v[k];
        }
        __syncthreads();
        if (t == 0) s_run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (t == 0) {
        const u32 total = s_run;
        const u32 nbase = lv.pass_b ? base - total : base + total;
        st->cur[bin] = nbase;
        if (bin == lv.c) { st->rng[lv.slot ^ 1u][0] = lv.pass_b ? nbase : base; st->rng[lv.slot ^ 1u][1] = lv.pass_b ? base : nbase; }
    }
}

// One tile of a level's stable scatter: row of source j's predecessor = first target row of (tile, byte) +- rank inside the tile.
// MODE 0 (three kernels per level, MSUFSORT_HIP_IND_CLASSIC): the first target rows come from k_ind_count + k_ind_scan (tile_hist).
// MODE 1 (short levels): the level is handled by ONE workgroup, tile after tile: the first target rows are the cursors themselves,
//                moved here; B* sources fetch their characters here too (no k_ind_count ran).
// MODE 2 (single pass): no count / scan kernels ran either; the first target rows come from a decoupled look-back over the
//                tile status words (status, epoch, ntiles, s_base = the cursors as they were when the level started).
template <int MODE>
__device__ __forceinline__ void ind_tile(IndState* st, const IndLevel& lv, const IndTables& tb, u32* sa, u32* pc, const u8* __restrict__ text,
                                         const u32* __restrict__ tile_hist, u32 tile, u32 lo, u32 hi, u32 my_code,
                                         u32 (*wcnt)[256], u32* goff, const u32* s_sub,
                                         u64* status = nullptr, u32 epoch = 0, u32 ntiles = 0, const u32* s_base = nullptr)
{
    constexpr bool SMALL = MODE == 1, FUSED = MODE == 2, OWN_STARS = MODE != 0;
    const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const u32 cnt = hi - lo;
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    __shared__ u32 s_fail;
    __shared__ u8 s_codes[256];            // dense number of every byte value (255: not in use; all 256 in use: the identity)
    __shared__ u8 s_syms[256];             // ... and back
    const u32 pbits = tb.pc_bits;          // (workgroup-uniform)
    __syncthreads();
    if (t == 0) s_fail = 0u;
    s_codes[t] = (u8)my_code;
    s_syms[t] = t < tb.nb ? tb.sym[t] : (u8)0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) wcnt[w2][t] = 0;
    if (MODE == 0) goff[t] = (my_code != 255u || tb.nb == 256u) ? tile_hist[(u64)tile * tb.nb + my_code] : 0u;
    __syncthreads();
    u32 j[IND_ITEMS], bin[IND_ITEMS], npc[IND_ITEMS];          // bin[i]: target byte (256: none) in bits 0..8, from the ranking on also the rank inside
                                                                 // the wave's share of the tile in bits 16.. (one register per source instead of two)
    u32 starmask = 0;
    // (phases, so that all of a thread's loads of one kind are in flight together: rows, then characters)
#pragma unroll
    for (int i = 0; i < IND_ITEMS; ++i) {
        const u32 q = tile * IND_TILE + wv * (IND_ITEMS * 64u) + i * 64u + lane;
        j[i] = 0; bin[i] = 256u; npc[i] = 0;
        if (q < cnt) j[i] = sa[lv.pass_b ? hi - 1u - q : lo + q];
    }
    // the characters in front of the sources: with the rows (pc[]); B* rows (level 0 of pass B) whose first-stage sort did not
    // leave them (PC_UNKNOWN, k_place_bstar) fetch them from the text here - and only those
#pragma unroll
    for (int i = 0; i < IND_ITEMS; ++i) {
        const u32 q = tile * IND_TILE + wv * (IND_ITEMS * 64u) + i * 64u + lane;
        if (q < cnt) npc[i] = pc[lv.pass_b ? hi - 1u - q : lo + q];
    }
    if (OWN_STARS && lv.stars) {
#pragma unroll
        for (int i = 0; i < IND_ITEMS; ++i) {
            const u32 q = tile * IND_TILE + wv * (IND_ITEMS * 64u) + i * 64u + lane;
            if (q < cnt && npc[i] == PC_UNKNOWN) { starmask |= 1u << i; npc[i] = ind_fetch_p(text, j[i], pbits, s_codes); }
        }
    }
#pragma unroll
    for (int i = 0; i < IND_ITEMS; ++i) {
        const u32 q = tile * IND_TILE + wv * (IND_ITEMS * 64u) + i * 64u + lane;
        if (q < cnt) {
            const u32 w = npc[i];
            if ((starmask >> i) & 1u) pc[hi - 1u - q] = w;
            bin[i] = ind_bin(lv, j[i], w);
            // the new row inherits my other characters: the next one as a byte again, the rest move down
            const u32 older = (w >> 8) & 0xffffu;
            if (pbits == 8u) npc[i] = older | (((w >> 24) - 1u) << 24);
            else npc[i] = (u32)s_syms[older & ((1u << pbits) - 1u)] | ((older >> pbits) << 8) | (((w >> 24) - 1u) << 24);
        }
    }
#pragma unroll
    for (int i = 0; i < IND_ITEMS; ++i)
        if (bin[i] < 256u && (npc[i] >> 24) == 0u) npc[i] = ind_fetch_p(text, j[i] - 1u, pbits, s_codes);                // ... or fetches its own
    const u32 nbits = tb.nsym > 1u ? 32u - (u32)__builtin_clz(tb.nsym - 1u) : 1u;     // bits of a dense byte number: 5 for a text, 3 for DNA
#pragma unroll
    for (int i = 0; i < IND_ITEMS; ++i) {
        // lanes of this row with my byte (rows of a wave are taken in order: stable): one ballot per bit of its dense number
        const bool on = bin[i] < 256u;
        const u32 cb = on ? (u32)s_codes[bin[i]] : 0u;
        u64 peers = __ballot(on);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if ((u32)b < nbits) {
                const bool bit = (cb >> b) & 1u;
                const u64 bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
        }
        if (on) {
            const int leader = __ffsll((long long)peers) - 1;
            u32 old = 0;
            if ((int)lane == leader) old = atomicAdd(&wcnt[wv][bin[i]], (u32)__popcll(peers));
            old = __shfl(old, leader, 64);
            bin[i] |= (old + (u32)__popcll(peers & lt_mask)) << 16;
        }
    }
    __syncthreads();
    u32 tot_t = 0;                         // what this tile writes for byte t
    {
        u32 o = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) { const u32 v = wcnt[w2][t]; wcnt[w2][t] = o; o += v; }
        tot_t = o;
        if (SMALL && o) {                  // claim the rows: the cursor of byte t moves by what this tile writes
            const u32 base = __atomic_load_n(&st->cur[t], __ATOMIC_RELAXED);
            goff[t] = lv.pass_b ? base - 1u : base;
            __atomic_store_n(&st->cur[t], lv.pass_b ? base - o : base + o, __ATOMIC_RELAXED);
        }
        if (FUSED && (my_code != 255u || tb.nb == 256u))        // publish this tile's counts first: the tiles behind it add them up
            __hip_atomic_store(status + (u64)tile * tb.nb + my_code, ind_status(tile == 0 ? IND_ST_INC : IND_ST_AGG, epoch, o), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (FUSED) goff[t] = 0;
    }
    if (FUSED) {
        // Decoupled look-back, eight lanes per byte value (32 values at a time): the lanes read the status words of the eight
        // tiles in front of this one and add up their counts back to the nearest running total (further back if there is
        // none among them).  Those tiles were handed out earlier, so their workgroups are running and publish without waiting
        // for anybody.
        __syncthreads();
        if (tile != 0) {
            const u32 sub = t & 7u, grp = lane >> 3;
            for (u32 cbase = 0; cbase < tb.nsym; cbase += 32u) {
                const u32 code = cbase + (t >> 3);
                const bool active = code < tb.nsym;
                u32 excl = 0, pbase = tile, spins = 0;
                const u32 spin_limit = st->spin_limit;
                bool poisoned = false;
                bool done = !active;
                for (;;) {
                    const bool in = !done && sub < pbase;
                    u64 v = 0;
                    if (in) v = __hip_atomic_load(status + (u64)(pbase - 1u - sub) * tb.nb + code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool ready = !in || ((((v >> 48) & 0x3fffu) == (epoch & 0x3fffu)) && (v >> 62) != 0);
                    const u64 inc_m = __ballot(in && ready && (v >> 62) == IND_ST_INC), wait_m = __ballot(!ready);
                    const u32 gi = (u32)(inc_m >> (grp * 8u)) & 255u, gw = (u32)(wait_m >> (grp * 8u)) & 255u;
                    const int f = gi ? __ffs((int)gi) - 1 : 7;                   // nearest running total, or the whole window
                    const bool blocked = (gw & ((2u << f) - 1u)) != 0;
                    poisoned |= in && ready && (v & IND_ST_POISON) != 0;
                    u32 part = (!blocked && in && (int)sub <= f) ? (u32)v : 0u;
                    part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
                    if (!done && !blocked) { excl += part; if (gi) done = true; else pbase -= 8u; }
                    if (!done && blocked && ++spins > spin_limit) { atomicOr(&st->flags, IND_FLAG_LOOKBACK); poisoned = true; done = true; }
                    if (__ballot(!done) == 0) break;
                    if (__ballot(blocked)) __builtin_amdgcn_s_sleep(1);
                }
                if (active && sub == 0) goff[tb.sym[code]] = excl;
                if (poisoned) s_fail = 1u;
            }
        }
        __syncthreads();
        if (my_code != 255u || tb.nb == 256u) {
            const u32 excl = goff[t];
            const u32 mine_tot = tot_t;
            if (tile != 0) __hip_atomic_store(status + (u64)tile * tb.nb + my_code, ind_status(IND_ST_INC, epoch, ((u64)excl + mine_tot) | (s_fail ? IND_ST_POISON : 0ull)),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 base = s_base[t];
            goff[t] = lv.pass_b ? base - 1u - excl : base + excl;
            if (tile == ntiles - 1u) {           // the last tile knows the level's totals: cursors move, the next level's rows are named
                const u32 nbase = lv.pass_b ? base - (excl + mine_tot) : base + (excl + mine_tot);
                st->cur[t] = nbase;
                if (t == lv.c) { st->rng[lv.slot ^ 1u][0] = lv.pass_b ? nbase : base; st->rng[lv.slot ^ 1u][1] = lv.pass_b ? base : nbase; }
            }
        }
    }
    // A look-back that timed out (this tile's, or one in front of it: the poison bit of the status words) leaves `excl` short:
    // nothing computed from it may be written.  The flag in the state is sticky - every later level and bucket returns at
    // once (ind_failed) and the host rebuilds with the sort-all path.  (s_fail is workgroup-uniform behind the barrier: the
    // caller's loop has barriers.)
    __syncthreads();
    if (FUSED && s_fail) return;
#pragma unroll
    for (int i = 0; i < IND_ITEMS; ++i)
        if ((bin[i] & 0x1ffu) < 256u) {
            const u32 b = bin[i] & 0x1ffu;
            const u32 k = wcnt[wv][b] + (bin[i] >> 16);
            const u32 dst = lv.pass_b ? goff[b] - k : goff[b] + k;
            sa[dst] = j[i] - 1u;
            pc[dst] = npc[i];
        }
}

__global__ __launch_bounds__(256) void k_ind_scatter(IndState* st, IndLevel lv, u32* sa, u32* pc,
                                                     const u8* __restrict__ text, IndTables tb, const u32* __restrict__ tile_hist)
{
    __shared__ u32 wcnt[4][256];
    __shared__ u32 goff[256];
    const u32 lo = st->rng[lv.slot][0], hi = st->rng[lv.slot][1];
    const u32 ntiles = (hi - lo + IND_TILE - 1) / IND_TILE;
    const u32 my_code = tb.code[threadIdx.x];
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
        ind_tile<0>(st, lv, tb, sa, pc, text, tile_hist, tile, lo, hi, my_code, wcnt, goff, nullptr);
}

// One level in ONE pass over its rows (instead of k_ind_count + k_ind_scan + k_ind_scatter): tiles are handed out in order by a
// ticket counter, every tile ranks its sources, publishes its per-byte counts and learns its first target rows from the tiles in
// front of it (decoupled look-back), then writes.  The rows are read once, the B* rows fetch their characters here.
#ifndef IND_WAVES
#define IND_WAVES 1
#endif
__global__ __launch_bounds__(256, IND_WAVES) void k_ind_fused(IndState* st, IndLevel lv, u32 epoch, u32* sa, u32* pc, const u8* __restrict__ text, IndTables tb,
                                                   u64* status)
{
    __shared__ u32 wcnt[4][256];
    __shared__ u32 goff[256];
    __shared__ u32 s_base[256];
    __shared__ u32 s_sub[257];
    __shared__ u32 s_tile;
    const u32 t = threadIdx.x;
    if (ind_failed(st)) return;          // an earlier level lost a look-back: its rows (and everything induced from them) are not there
    s_base[t] = st->cur[t];              // the cursors as the level finds them (the last tile moves them: read before taking a tile)
    const u32 lo = st->rng[lv.slot][0], hi = st->rng[lv.slot][1];
    const u32 ntiles = (hi - lo + IND_TILE - 1) / IND_TILE;
    if (lv.stars) { s_sub[t] = tb.sub_start[lv.c * 256u + t]; if (t == 0) s_sub[256] = tb.bkt[lv.c + 1]; }
    const u32 my_code = tb.code[t];
    if (ntiles == 0) {                   // nothing to read: the next level is empty as well
        if (blockIdx.x == 0 && t == 0) { st->rng[lv.slot ^ 1u][0] = 0; st->rng[lv.slot ^ 1u][1] = 0; }
        return;
    }
    // Hand-out of the tiles.  Every workgroup draws ONE ticket when it starts: its arrival order a, which is also its first tile -
    // a tile is only ever given to a RUNNING workgroup and in increasing order, so the tiles a look-back waits for are always
    // being worked on.  What happens behind the first round is settled once per launch (st->mode): when a workgroup comes
    // back from its first tile and the counter shows that ALL G workgroups of the launch have drawn (= are running, nobody
    // waits for a free CU), it proposes FIXED: workgroup a takes a + G, a + 2G, ... with no further atomics - one memory
    // round trip less in front of every tile's published counts (-6 .. -11 % per level, measured in round 3 with an
    // environment switch that TRUSTED the launch to be resident; this handshake observes it).  If some workgroup has not
    // started by then (a grid larger than the chip holds, a GPU shared with another queue) it proposes TICKET: everybody keeps
    // drawing from the counter, as before.  One compare-and-swap decides; FIXED can only win while nobody has drawn twice.
    const u32 e = epoch & (IND_MAX_LEVELS - 1u);
    const u32 G = gridDim.x;
    if (t == 0) s_tile = atomicAdd(&st->ticket[e], 1u);
    __syncthreads();
    const u32 a = s_tile;
    if (a >= ntiles) return;
    ind_tile<2>(st, lv, tb, sa, pc, text, nullptr, a, lo, hi, my_code, wcnt, goff, s_sub, status, epoch, ntiles, s_base);
    if (ntiles <= G) {                        // every tile is somebody's first: nothing to settle
        if (a + G >= ntiles) return;
    }
    __syncthreads();
    if (t == 0) {
        u32 m = __hip_atomic_load(&st->mode[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (u32 spin = 0; m == 0u; ++spin) {
            const u32 drawn = __hip_atomic_load(&st->ticket[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 want = drawn == G ? IND_MODE_FIXED : (spin >= 32u ? IND_MODE_TICKET : 0u);      // (stragglers get a few microseconds)
            if (want) {
                u32 expect = 0u;
                if (__hip_atomic_compare_exchange_strong(&st->mode[e], &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    atomicAdd(&st->decided[want == IND_MODE_FIXED ? 0 : 1], 1u);
            }
            else __builtin_amdgcn_s_sleep(8);
            m = __hip_atomic_load(&st->mode[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_tile = m;
    }
    __syncthreads();
    const u32 mode = s_tile;
    if (mode == IND_MODE_FIXED) {
        // all G workgroups run, each takes its tiles in increasing order: the smallest unfinished tile is always being worked on
        for (u32 tile = a + G; tile < ntiles; tile += G) {
            __syncthreads();
            ind_tile<2>(st, lv, tb, sa, pc, text, nullptr, tile, lo, hi, my_code, wcnt, goff, s_sub, status, epoch, ntiles, s_base);
        }
        return;
    }
    for (;;) {
        __syncthreads();
        if (t == 0) s_tile = atomicAdd(&st->ticket[e], 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= ntiles) break;
        ind_tile<2>(st, lv, tb, sa, pc, text, nullptr, tile, lo, hi, my_code, wcnt, goff, s_sub, status, epoch, ntiles, s_base);
    }
}

// Levels that are known to be short (at most one tile, from the run-length counts of k_types): ONE workgroup takes all the
// remaining levels of a bucket, one after the other - most levels of a text are a handful of rows, and a level of three
// launches costs more than it computes.
__global__ __launch_bounds__(256) void k_ind_small(IndState* st, IndLevel lv, u32 nlevels, u32* sa, u32* pc, const u8* __restrict__ text, IndTables tb)
{
    __shared__ u32 wcnt[4][256];
    __shared__ u32 goff[256];
    __shared__ u32 s_sub[257];
    const u32 t = threadIdx.x;
    if (ind_failed(st)) return;
    if (lv.stars) { s_sub[t] = tb.sub_start[lv.c * 256u + t]; if (t == 0) s_sub[256] = tb.bkt[lv.c + 1]; }
    const u32 my_code = tb.code[t];
    u32 slot = lv.slot;
    for (u32 l = 0; l < nlevels; ++l) {
        IndLevel cur = lv;
        cur.slot = slot; cur.stars = l == 0 ? lv.stars : 0u;
        __syncthreads();
        const u32 lo = __atomic_load_n(&st->rng[slot][0], __ATOMIC_RELAXED), hi = __atomic_load_n(&st->rng[slot][1], __ATOMIC_RELAXED);
        const u32 before = __atomic_load_n(&st->cur[lv.c], __ATOMIC_RELAXED);
        if (hi == lo) break;
        const u32 ntiles = (hi - lo + IND_TILE - 1) / IND_TILE;
        for (u32 tile = 0; tile < ntiles; ++tile) ind_tile<1>(st, cur, tb, sa, pc, text, nullptr, tile, lo, hi, my_code, wcnt, goff, s_sub);
        __threadfence();
        __syncthreads();
        if (t == 0) {
            const u32 after = __atomic_load_n(&st->cur[lv.c], __ATOMIC_RELAXED);
            __atomic_store_n(&st->rng[slot ^ 1u][0], lv.pass_b ? after : before, __ATOMIC_RELAXED);
            __atomic_store_n(&st->rng[slot ^ 1u][1], lv.pass_b ? before : after, __ATOMIC_RELAXED);
        }
        slot ^= 1u;
    }
}

// after a pass: every cursor must have arrived where the histograms said it would
__global__ __launch_bounds__(256) void k_ind_check(IndState* __restrict__ st, IndTables tb, u32 kind, u32 c)
{
    const u32 t = threadIdx.x;
    bool bad;
    if (kind == 0) bad = t <= c && st->cur[t] != tb.sub_start[t * 256u + c] + tb.sub_bs[t * 256u + c];      // pass B, after bucket c
    else bad = st->cur[t] != tb.aend[t];                                                                    // after pass A
    if (bad) atomicOr(&st->flags, kind == 0 ? IND_FLAG_CURSOR : IND_FLAG_CURSOR_A);
}
