// BWT / LCP / validation kernels (forward BWT: reference msufsort.cpp:1771-1817 semantics via
// BWT[r] = T[SA[r]-1]; LCP: reference src/executable/msufsort/main.cpp:16-159; checker: main.cpp:210-270).
#pragma once
#include "sa_kernels.hip.h"

// row r with SA[r] == 0 (the sentinel row, cpp:1283-1286)
template <bool W>
__global__ __launch_bounds__(256) void k_find_sentinel(const typename Wd<W>::sa_t* __restrict__ sa, u64 rows, unsigned long long* __restrict__ sent)
{
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x; r < rows; r += (u64)gridDim.x * 256u)
        if (sa[r] == 0) *sent = r;
}

// BWT bytes with the sentinel row removed (cpp:1811-1815)
template <bool W>
__global__ __launch_bounds__(256) void k_bwt_gather(const u8* __restrict__ text, const typename Wd<W>::sa_t* __restrict__ sa, u64 rows,
                                                    const unsigned long long* __restrict__ sentp, u8* __restrict__ out)
{
    const u64 sent = *sentp;
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x; r < rows; r += (u64)gridDim.x * 256u) {
        const u64 v = sa[r];
        if (v != 0) out[r - (r > sent)] = text[v - 1];
    }
}

// Two-stage builds (induce_kernels.hip.h) leave the character in front of every suffix next to its row: the BWT is a
// sequential pass over those instead of n random text reads - how the reference's own forward transform takes its output
// out of the second stage (cpp:1061-1492) rather than from a finished suffix array.
__global__ __launch_bounds__(256) void k_bwt_from_pc(const u8* __restrict__ text, const u32* __restrict__ sa, const u32* __restrict__ pc, u64 rows,
                                                     const unsigned long long* __restrict__ sentp, u8* __restrict__ out)
{
    const u64 sent = *sentp;
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x; r < rows; r += (u64)gridDim.x * 256u) {
        const u32 v = sa[r];
        if (v != 0) out[r - (r > sent)] = r == 0 ? text[v - 1] : (u8)pc[r];      // (row 0 is the empty suffix: no characters were kept for it)
    }
}


// demo convention (main.cpp:66-101): out[i] = lcp(SA[i+1], SA[i+2]), i in [0, n-2]; out[n-1] = 0.
// Direct compare like the demo's match_length, but capped: a pair that is still equal after `cap` bytes raises
// *flag and the host switches to the PLCP method below (periodic inputs have LCPs of 10^4..10^5).
__global__ __launch_bounds__(256) void k_lcp(const u8* __restrict__ text, u64 n, const u32* __restrict__ sa, u32* __restrict__ out,
                                             u32 cap, u32* __restrict__ flag)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u) {
        if (*reinterpret_cast<volatile u32*>(flag)) return;       // somebody met a pair beyond the cap: the host switches to PLCP, stop here
        u32 v = 0;
        if (i + 1 < n) {
            u64 a = sa[i + 1], b = sa[i + 2];
            if (a > b) { const u64 x = a; a = b; b = x; }
            u64 m = 0;
            bool open = true;
            while (open && b + m + 8 <= n && m < cap) {
                u64 x, y;
                __builtin_memcpy(&x, text + a + m, 8);
                __builtin_memcpy(&y, text + b + m, 8);
                if (x != y) { m += (u64)(__ffsll((long long)(x ^ y)) - 1) >> 3; open = false; }
                else m += 8;
            }
            if (open) {
                if (m >= cap) *flag = 1u;
                while (b + m < n && text[a + m] == text[b + m] && m < (u64)cap + 16) ++m;
            }
            v = (u32)m;
        }
        out[i] = v;
    }
}

// PLCP method (Karkkainen-Manzini-Puglisi "permuted LCP"): phi[SA[r]] = SA[r-1]; PLCP[i] = lcp(i, phi[i]) obeys
// PLCP[i] >= PLCP[i-1] - 1, so a thread that walks a chunk of consecutive text positions only pays the
// full match once.  LCP[r] = PLCP[SA[r]].
#define PLCP_CHUNK 128u
__global__ __launch_bounds__(256) void k_phi(const u32* __restrict__ sa, u64 n, u32* __restrict__ phi)
{
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x + 1; r <= n; r += (u64)gridDim.x * 256u) phi[sa[r]] = sa[r - 1];
}

__global__ __launch_bounds__(256) void k_plcp(const u8* __restrict__ text, u64 n, const u32* __restrict__ phi, u32* __restrict__ plcp)
{
    const u64 nchunks = (n + PLCP_CHUNK - 1) / PLCP_CHUNK;
    for (u64 c = (u64)blockIdx.x * 256u + threadIdx.x; c < nchunks; c += (u64)gridDim.x * 256u) {
        u64 l = 0;
        const u64 end = (c + 1) * PLCP_CHUNK < n ? (c + 1) * PLCP_CHUNK : n;
        for (u64 i = c * PLCP_CHUNK; i < end; ++i) {
            const u64 j = phi[i];
            if (j >= n) { plcp[i] = 0; l = 0; continue; }        // predecessor is the empty suffix
            l = l ? l - 1 : 0;
            const u64 hi = i > j ? i : j;
            bool open = true;
            while (open && hi + l + 8 <= n) {
                u64 x, y;
                __builtin_memcpy(&x, text + i + l, 8);
                __builtin_memcpy(&y, text + j + l, 8);
                if (x != y) { l += (u64)(__ffsll((long long)(x ^ y)) - 1) >> 3; open = false; }
                else l += 8;
            }
            if (open) while (hi + l < n && text[i + l] == text[j + l]) ++l;
            plcp[i] = (u32)l;
        }
    }
}

__global__ __launch_bounds__(256) void k_lcp_from_plcp(const u32* __restrict__ sa, u64 n, const u32* __restrict__ plcp, u32* __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u)
        out[i] = (i + 1 < n) ? plcp[sa[i + 2]] : 0u;
}

// validate_suffix_array (main.cpp:236-270): SA[0] == n, adjacent suffixes strictly increasing ("shorter is
// smaller", main.cpp:210-232), plus range and permutation checks.  The demo compares adjacent suffixes byte by
// byte, which is O(n * LCP) and takes minutes on periodic inputs; the order test used here is the exact linear one:
// with rank = inverse permutation (rank of the empty suffix = 0),
//     suffix a < suffix b  <=>  T[a] < T[b]  or  (T[a] == T[b] and rank[a+1] < rank[b+1]).
template <bool W>
__global__ __launch_bounds__(256) void k_validate_perm(u64 n, const typename Wd<W>::sa_t* __restrict__ sa, typename Wd<W>::sa_t* __restrict__ rank /* n+1 */,
                                                       unsigned long long* __restrict__ errors)
{
    typedef typename Wd<W>::sa_t sa_t;
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x; r <= n; r += (u64)gridDim.x * 256u) {
        const sa_t v = sa[r];
        if (r == 0) { if (v != (sa_t)n) atomicAdd(errors, 1ull); rank[n] = 0; continue; }
        if (v >= n) { atomicAdd(errors, 1ull); continue; }
        rank[v] = (sa_t)r;                      // duplicates: one row wins, k_validate_order sees the others
    }
}

template <bool W>
__global__ __launch_bounds__(256) void k_validate_order(const u8* __restrict__ text, u64 n, const typename Wd<W>::sa_t* __restrict__ sa,
                                                        const typename Wd<W>::sa_t* __restrict__ rank, unsigned long long* __restrict__ errors)
{
    typedef typename Wd<W>::sa_t sa_t;
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x + 1; r <= n; r += (u64)gridDim.x * 256u) {
        const sa_t b = sa[r];
        if (b >= n) continue;                               // already counted by k_validate_perm
        if (rank[b] != (sa_t)r) { atomicAdd(errors, 1ull); continue; }      // permutation: every value names its own row
        if (r < 2) continue;
        const sa_t a = sa[r - 1];
        if (a >= n) continue;
        const u32 ca = text[a], cb = text[b];
        const bool ok = ca < cb || (ca == cb && rank[a + 1] < rank[b + 1]);
        if (!ok) atomicAdd(errors, 1ull);
    }
}

// ================================================================================================
// Inverse BWT (reverse_burrows_wheeler_transform, reference msufsort.cpp:1821-2096).
//   rows 0..n of the BWT matrix; byte i of the stored BWT is row i + (i >= sentinel) (cpp:1907-1913).
//   (i)   per-tile symbol histogram                      (cpp:1842-1879)
//   (ii)  device-wide exclusive scan, symbol-major / tile-minor, base 1   (cpp:1880-1889)
//   (iii) stable ranked scatter -> forward links link[k] = row            (cpp:1891-1919)
//   (iv)  marker-terminated chain walks, one chain per lane               (cpp:1922-2063)
//   (v)   fragment order by list ranking (pointer jumping) instead of the serial stitch (cpp:2065-2095)
// ================================================================================================
#define IBWT_WT 8192u          // rows per wave-tile
#ifndef IBWT_S
#define IBWT_S 512u            // splitter stride (rows that are multiples of S start a chain)
#endif

__device__ __forceinline__ u32 ibwt_sym(const u8* __restrict__ bwt, u32 row, u32 sent)
{
    return bwt[row - (row > sent)];
}

__global__ __launch_bounds__(256) void k_ibwt_count(const u8* __restrict__ bwt, u32 rows, u32 sent, u32 ntiles, u32* __restrict__ counts)
{
    // eight copies of every wave's 256 bins (copy = lane & 7): text puts a third of its bytes on a handful of symbols, and
    // lanes that meet on one LDS address are served one after the other
    __shared__ u32 h[4][8][256];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 tile = blockIdx.x * 4 + w;
    for (u32 i = lane; i < 8u * 256u; i += 64) (&h[w][0][0])[i] = 0;
    __syncthreads();
    if (tile < ntiles) {
        const u32 beg = tile * IBWT_WT;
        const u32 end = (rows - beg < IBWT_WT) ? rows : beg + IBWT_WT;
        // the rows [beg, end) without the sentinel row are the stored bytes [b0, b1): 16 of them per lane and load
        const u32 b0 = beg - (beg > sent ? 1u : 0u), b1 = end - (end > sent ? 1u : 0u);
        for (u32 o = b0 + lane * 16u; o < b1; o += 64u * 16u) {
            u32 v[4] = {0, 0, 0, 0};
            const u32 lim = b1 - o >= 16u ? 16u : b1 - o;
            if (lim == 16u) __builtin_memcpy(v, bwt + o, 16);
            else for (u32 k = 0; k < lim; ++k) v[k >> 2] |= (u32)bwt[o + k] << (8u * (k & 3u));
#pragma unroll
            for (u32 k = 0; k < 16; ++k) if (k < lim) atomicAdd(&h[w][lane & 7u][(v[k >> 2] >> (8u * (k & 3u))) & 255u], 1u);
        }
    }
    __syncthreads();
    if (tile < ntiles)
        for (u32 i = lane; i < 256; i += 64) {
            u32 sum = 0;
#pragma unroll
            for (u32 q = 0; q < 8; ++q) sum += h[w][q][i];
            counts[(u64)i * ntiles + tile] = sum;
        }
}

// generic device-wide exclusive scan of u32[N]: partial -> top -> final
#define SCAN_ITEMS 8
#define SCAN_BLOCK (1024 * SCAN_ITEMS)
__device__ __forceinline__ u32 block_excl_scan_1024(u32 v, u32* wsum /*16*/, u32& block_total)
{
    u32 wt;
    const u32 e = wave_excl_scan(v, wt);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = wt;
    __syncthreads();
    u32 wbase = 0, tot = 0;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) { const u32 s = wsum[k]; if (k < (threadIdx.x >> 6)) wbase += s; tot += s; }
    __syncthreads();
    block_total = tot;
    return wbase + e;
}

__global__ __launch_bounds__(1024) void k_scan_partial(const u32* __restrict__ in, u64 N, u32* __restrict__ block_sums)
{
    __shared__ u32 wsum[16];
    const u64 base = (u64)blockIdx.x * SCAN_BLOCK + (u64)threadIdx.x * SCAN_ITEMS;
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < N) s += in[base + k];
    u32 tot;
    (void)block_excl_scan_1024(s, wsum, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_scan_top(u32* __restrict__ sums, u32 nb, u32 init)
{
    __shared__ u32 wsum[16];
    __shared__ u32 s_carry;
    if (threadIdx.x == 0) s_carry = init;
    __syncthreads();
    for (u32 b = 0; b < nb; b += 1024u) {
        const u32 i = b + threadIdx.x;
        const u32 v = i < nb ? sums[i] : 0u;
        u32 tot;
        const u32 e = block_excl_scan_1024(v, wsum, tot);
        const u32 carry = s_carry;
        if (i < nb) sums[i] = carry + e;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_scan_final(u32* __restrict__ data, u64 N, const u32* __restrict__ block_sums)
{
    __shared__ u32 wsum[16];
    const u64 base = (u64)blockIdx.x * SCAN_BLOCK + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < N) ? data[base + k] : 0u; s += v[k]; }
    u32 tot;
    u32 e = block_excl_scan_1024(s, wsum, tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < N) data[base + k] = e; e += v[k]; }
}

// stable ranked scatter: link[C[c] + rank] = row  (one wave per tile, 64 rows per step)
__global__ __launch_bounds__(256) void k_ibwt_scatter(const u8* __restrict__ bwt, u32 rows, u32 sent, u32 ntiles,
                                                      const u32* __restrict__ offs, u32* __restrict__ link)
{
    __shared__ u32 cur[4][256];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 tile = blockIdx.x * 4 + w;
    if (blockIdx.x == 0 && threadIdx.x == 0) link[0] = sent;      // cpp:1891
    if (tile < ntiles) for (u32 i = lane; i < 256; i += 64) cur[w][i] = offs[(u64)i * ntiles + tile];
    __syncthreads();
    if (tile >= ntiles) return;
    const u32 beg = tile * IBWT_WT;
    const u32 end = (rows - beg < IBWT_WT) ? rows : beg + IBWT_WT;
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (u32 rb = beg; rb < end; rb += 64u * 4u) {               // four byte loads in flight per lane
        u32 cs[4];
#pragma unroll
        for (u32 k = 0; k < 4; ++k) { const u32 r = rb + 64u * k + lane; cs[k] = (r < end && r != sent) ? ibwt_sym(bwt, r, sent) : 0u; }
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
        const u32 r0 = rb + 64u * k;
        if (r0 >= end) break;
        const u32 r = r0 + lane;
        const bool valid = r < end && r != sent;
        const u32 c = cs[k];
        u64 mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (c >> b) & 1u;
            const u64 bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        if (valid) {
            const int leader = __ffsll((long long)mask) - 1;
            u32 old = 0;
            if ((int)lane == leader) old = atomicAdd(&cur[w][c], (u32)__popcll(mask));
            old = __shfl(old, leader, 64);
            link[old + (u32)__popcll(mask & lt_mask)] = r;
        }
        }
    }
}

__device__ __forceinline__ bool ibwt_marked(u32 row, u32 sent) { return (row & (IBWT_S - 1)) == 0 || row == sent; }
__device__ __forceinline__ u32 ibwt_id(u32 row, u32 sent, u32 kreg) { return (row == sent && (sent & (IBWT_S - 1))) ? kreg : row / IBWT_S; }
__device__ __forceinline__ u32 ibwt_start(u32 id, u32 sent, u32 kreg) { return id == kreg ? sent : id * IBWT_S; }

// ONE walk: lanes pull chains from a queue and store the text bytes they meet into the chain's own fixed-size
// buffer (sequential per chain, so L2 merges the byte stores).  A chain that reaches IBWT_CW hops without meeting a
// marker row is cut there: the same lane continues into a freshly numbered segment (ids >= K) and links the two.
// Afterwards nxt/dist describe a linked list of segments of at most IBWT_CW bytes each; list ranking gives every
// segment its text position and k_ibwt_assemble copies the buffers out with coalesced stores - n dependent
// random reads in total instead of the 2n of a count-then-write scheme.
//
// A hop is ONE 4-byte load from the link table (4(n+1) bytes; the reference packs link + symbol in 5, cpp:1826-1836):
// the symbol of text position p is F[row(p)], and F is the sorted column - a search in the cumulative symbol counts C[],
// which live in LDS (a 4096-entry coarse table gives the first candidate, then a step or two), computed while the load
// of the next row is in flight.  Every lane runs IBWT_NCH independent chains (the reference interleaves many chains per
// thread the same way, cpp:1922,1976-2018): the walk is bound by dependent random sector reads, and the loads in
// flight per lane are what one lane can add to the memory-level parallelism.
#ifndef IBWT_CW
#define IBWT_CW 1024u
#endif
#ifndef IBWT_NCH
#define IBWT_NCH 2
#endif
__global__ __launch_bounds__(256) void k_ibwt_walk(const u32* __restrict__ link, const u32* __restrict__ offs /* [256][ntiles]: C[c] = offs[c * ntiles] */,
                                                   u32 ntiles, u32 rows, u32 sent, u32 kreg, u32 K, u32 kt_cap,
                                                   u32* __restrict__ queue /* [1] dynamic count, [2] overflow flag */,
                                                   u32* __restrict__ nxt, u32* __restrict__ dist, u8* __restrict__ segbuf)
{
    // chain ids 1 .. K-1 are dealt to the workgroups in contiguous shares and pulled from an LDS counter: one global
    // counter for all 4 M pulls of a 1 GiB input saturates (~90 returning atomics per us) and was what the walk waited
    // for; chain lengths are i.i.d., so a share of a few thousand chains is balanced to a few per cent
    __shared__ u32 s_next, s_end, s_shift;
    __shared__ u32 s_C[258];
    __shared__ u8 s_T[4096];
    const u32 t = threadIdx.x;
    s_C[t] = offs[(u64)t * ntiles];
    if (t == 0) {
        s_C[256] = rows; s_C[257] = 0xffffffffu;
        const u32 per = (K - 1u + gridDim.x - 1u) / gridDim.x;
        const u64 b = 1ull + (u64)blockIdx.x * per;
        s_next = b < K ? (u32)b : K;
        s_end = b + per < K ? (u32)(b + per) : K;
        u32 sh = 0;
        while (sh < 32 && ((rows - 1) >> sh) >= 4096u) ++sh;
        s_shift = sh;
    }
    __syncthreads();
    {   // coarse table: T[b] = the symbol whose rows contain row b << shift (largest c with C[c] <= that row; 0 below C[0])
        const u32 sh = s_shift;
        for (u32 b = t; b < 4096u; b += 256u) {
            const u64 row = (u64)b << sh;
            u32 lo = 0, hi = 256;                                   // invariant: C[lo] <= row (or lo == 0), C[hi] > row
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if ((u64)s_C[mid] <= row) lo = mid; else hi = mid; }
            s_T[b] = (u8)lo;
        }
    }
    __syncthreads();
    const u32 shift = s_shift;
#define IBWT_PULL() ([&]() { const u32 i_ = atomicAdd(&s_next, 1u); return i_ < s_end ? i_ : K; }())
    u32 id[IBWT_NCH], cur[IBWT_NCH], len[IBWT_NCH], my[IBWT_NCH];
    u64 acc[IBWT_NCH][4];              // the last (len & 31) bytes met: bytes leave as aligned 32-byte pieces (one scattered
                                       // write per 32 hops and chain: the walk is bound by random DRAM accesses, and every
                                       // 8-byte store of round 1's version was one more of them - 31 -> 28 ms at 1 GiB)
#pragma unroll
    for (int c = 0; c < IBWT_NCH; ++c) {
        id[c] = IBWT_PULL(); my[c] = id[c]; len[c] = 0; cur[c] = id[c] < K ? ibwt_start(id[c], sent, kreg) : 0u;
        acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0;
    }
#define IBWT_FLUSH(c, at) do { uint4* o_ = reinterpret_cast<uint4*>(segbuf + (u64)my[c] * IBWT_CW + (at)); \
        o_[0] = make_uint4((u32)acc[c][0], (u32)(acc[c][0] >> 32), (u32)acc[c][1], (u32)(acc[c][1] >> 32)); \
        o_[1] = make_uint4((u32)acc[c][2], (u32)(acc[c][2] >> 32), (u32)acc[c][3], (u32)(acc[c][3] >> 32)); \
        acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0; } while (0)
    for (;;) {
        bool any = false;
        u32 nx[IBWT_NCH];
#pragma unroll
        for (int c = 0; c < IBWT_NCH; ++c) { nx[c] = 0; if (id[c] < K) { nx[c] = link[cur[c]]; any = true; } }     // the dependent loads, all in flight together
        if (!any) break;
#pragma unroll
        for (int c = 0; c < IBWT_NCH; ++c) {
            if (id[c] >= K) continue;
            // symbol of the row I am leaving (independent of the load above)
            u32 sy = s_T[cur[c] >> shift];
            while (cur[c] >= s_C[sy + 1]) ++sy;
            {
                const u64 v = (u64)sy << (8u * (len[c] & 7u));
                const u32 wsel = (len[c] >> 3) & 3u;
                acc[c][0] |= wsel == 0u ? v : 0ull; acc[c][1] |= wsel == 1u ? v : 0ull;
                acc[c][2] |= wsel == 2u ? v : 0ull; acc[c][3] |= wsel == 3u ? v : 0ull;
            }
            ++len[c];
            if ((len[c] & 31u) == 0) IBWT_FLUSH(c, len[c] - 32u);
            const u32 r = nx[c];
            if (ibwt_marked(r, sent)) {
                if (len[c] & 31u) IBWT_FLUSH(c, len[c] & ~31u);      // (tail bytes beyond len are never read; the buffer has room: len < IBWT_CW here)
                nxt[my[c]] = ibwt_id(r, sent, kreg);
                dist[my[c]] = len[c];
                id[c] = IBWT_PULL();
                my[c] = id[c]; len[c] = 0;
                cur[c] = id[c] < K ? ibwt_start(id[c], sent, kreg) : 0u;
            } else {
                if (len[c] == IBWT_CW) {      // (a multiple of 32: acc has just been stored)
                    const u32 fresh = K + atomicAdd(&queue[1], 1u);
                    if (fresh >= kt_cap) { queue[2] = 1u; return; }          // cannot happen (capacity covers every cut)
                    nxt[my[c]] = fresh; dist[my[c]] = len[c];
                    my[c] = fresh; len[c] = 0;
                }
                cur[c] = r;
            }
        }
    }
#undef IBWT_PULL
#undef IBWT_FLUSH
}

// out[pos .. pos + len) = segment buffer; one wave per segment, 64 contiguous bytes per store instruction
__global__ __launch_bounds__(256) void k_ibwt_assemble(const u8* __restrict__ segbuf, const u32* __restrict__ len0,
                                                       const u32* __restrict__ dist_to_end, u32 KT, u32 n, u8* __restrict__ out)
{
    const u32 seg = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= KT || seg == 0) return;
    const u32 len = len0[seg];
    const u64 pos = (u64)n - dist_to_end[seg];
    const u8* src = segbuf + (u64)seg * IBWT_CW;
    // 16 bytes per lane (the buffer is aligned, the text position is not: unaligned 16-byte stores), bytes for the tail
    const u32 full = len & ~15u;
    for (u32 o = lane * 16u; o < full; o += 64u * 16u) {
        uint4 v;
        __builtin_memcpy(&v, src + o, 16);
        __builtin_memcpy(out + pos + o, &v, 16);
    }
    if (lane < (len & 15u)) out[pos + full + lane] = src[full + lane];
}

__global__ __launch_bounds__(256) void k_ibwt_jump(const u32* __restrict__ nxt, const u32* __restrict__ dist, u32 K,
                                                   u32* __restrict__ nxt2, u32* __restrict__ dist2)
{
    const u32 j = blockIdx.x * 256u + threadIdx.x;
    if (j >= K) return;
    const u32 x = nxt[j];
    dist2[j] = dist[j] + dist[x];
    nxt2[j] = nxt[x];
}

