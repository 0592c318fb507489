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

// Sharded forward transform ("for BWT gather n/G-byte slices instead"): the byte in front of every suffix of ONE
// finished slice, one byte per row (the sentinel row - the row of suffix 0 - gets a placeholder and is reported; the caller
// closes the hole once every rank knows where it is).  rows = hi - lo, row_lo = lo.
template <bool W>
__global__ __launch_bounds__(256) void k_bwt_slice(const u8* __restrict__ text, const typename Wd<W>::sa_t* __restrict__ sa_slice, u64 rows, u64 row_lo,
                                                   unsigned long long* __restrict__ sent, u8* __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < rows; i += (u64)gridDim.x * 256u) {
        const u64 v = sa_slice[i];
        out[i] = v ? text[v - 1] : (u8)0;
        if (v == 0) *sent = row_lo + i;
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

// The same for ONE region of rows [lo, hi) that has become final while the build goes on (two-stage build, engine.hip BwtRide).
// mode 0: the row of suffix 0 lies behind the region, 1: in front of it, 2: inside (k_find_row0 has written it to *sentp).
__global__ __launch_bounds__(256) void k_find_row0(const u32* __restrict__ sa, u64 lo, u64 hi, unsigned long long* __restrict__ sent)
{
    for (u64 r = lo + (u64)blockIdx.x * 256u + threadIdx.x; r < hi; r += (u64)gridDim.x * 256u)
        if (sa[r] == 0) *sent = r;
}
__global__ __launch_bounds__(256) void k_bwt_region(const u8* __restrict__ text, const u32* __restrict__ sa, const u32* __restrict__ pc, u64 lo, u64 hi, u32 mode,
                                                    const unsigned long long* __restrict__ sentp, u8* __restrict__ out)
{
    const u64 sent = mode == 2u ? *sentp : (mode == 1u ? 0ull : ~0ull);
    for (u64 r = lo + (u64)blockIdx.x * 256u + threadIdx.x; r < hi; r += (u64)gridDim.x * 256u) {
        const u32 v = sa[r];
        if (v != 0) out[r - (r > sent)] = r == 0 ? text[v - 1] : (u8)pc[r];
    }
}

// demo convention (main.cpp:66-101): out[i] = lcp(SA[i+1], SA[i+2]), i in [0, n-2]; out[n-1] = 0.
// Direct compare like the demo's match_length, but capped: a pair that is still equal after `cap` bytes raises
// *flag and the host switches to the PLCP method below (periodic inputs have LCPs of 10^4..10^5).
// the first 32 bytes of suffix p as four little-endian words, zero beyond the end of the text (no padding is assumed)
__device__ __forceinline__ void lcp_head(const u8* __restrict__ text, u64 n, u64 p, u64 (&w)[4])
{
    if (p + 32 <= n) {
        uint4 lo, hi;
        __builtin_memcpy(&lo, text + p, 16);
        __builtin_memcpy(&hi, text + p + 16, 16);
        w[0] = (u64)lo.x | ((u64)lo.y << 32); w[1] = (u64)lo.z | ((u64)lo.w << 32);
        w[2] = (u64)hi.x | ((u64)hi.y << 32); w[3] = (u64)hi.z | ((u64)hi.w << 32);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u64 x = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const u64 at = p + 8u * k + q; if (at < n) x |= (u64)text[at] << (8 * q); }
            w[k] = x;
        }
    }
}

// out[i] = lcp(SA[i+1], SA[i+2]).  A wave takes 63 consecutive rows: lane l fetches the first 32 bytes of suffix SA[base + l + 1]
// ONCE (two 16-byte loads from one line) and gets its right neighbour's from lane l + 1 - every suffix is touched by one lane
// instead of two, and pairs that differ inside 32 bytes (nearly all of a text, all of random bytes) need no second trip to
// memory; the bound of this kernel is random line fetches, one per row.  Longer matches continue 8 bytes at a time.
__global__ __launch_bounds__(256) void k_lcp(const u8* __restrict__ text, u64 n, const u32* __restrict__ sa, u32* __restrict__ out,
                                             u32 cap, u32* __restrict__ flag)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 nwin = (n + 62) / 63;
    for (u64 win = (u64)blockIdx.x * 4u + (threadIdx.x >> 6); win < nwin; win += (u64)gridDim.x * 4u) {
        if (*reinterpret_cast<volatile u32*>(flag)) return;       // somebody met a pair beyond the cap: the host switches to PLCP, stop here
        const u64 i = win * 63 + lane;                            // my suffix is SA[i + 1]; my row (lanes 0..62) is i
        const bool have = i < n;                                  // rows 1 .. n of the suffix array exist
        const u64 a0 = have ? sa[i + 1] : 0;
        u64 w[4] = {0, 0, 0, 0};
        if (have) lcp_head(text, n, a0, w);
        const u64 b0 = __shfl_down(a0, 1, 64);
        u64 v4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v4[k] = __shfl_down(w[k], 1, 64);
        if (lane == 63u || i + 1 >= n) {                          // no row here: the last suffix's row is 0 by definition
            if (lane != 63u && have) out[i] = 0;
            continue;
        }
        u64 a = a0, b = b0;
        if (a > b) { const u64 x = a; a = b; b = x; }
        const u64 limit = n - b;                                  // bytes of the shorter suffix
        u64 m = 32;
        bool open = true;
#pragma unroll
        for (int k = 3; k >= 0; --k) { const u64 x = w[k] ^ v4[k]; if (x) { m = 8u * k + ((u64)(__ffsll((long long)x) - 1) >> 3); open = false; } }
        if (m >= limit) { m = limit; open = false; }              // (zero fill beyond the end compares equal)
        if (open) {
            while (open && b + m + 8 <= n && m < cap) {
                u64 x, y;
                __builtin_memcpy(&x, text + a + m, 8);
                __builtin_memcpy(&y, text + b + m, 8);
                if (x != y) { m += (u64)(__ffsll((long long)x ^ (long long)y) - 1) >> 3; open = false; }
                else m += 8;
            }
            if (open) while (b + m < n && text[a + m] == text[b + m] && m < (u64)cap + 16) ++m;
        }
        if (m >= cap) *flag = 1u;
        out[i] = (u32)m;
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
    // lanes that meet on one LDS address are served one after the other.  The copies of one symbol sit next to each other
    // (bin * 8 + copy): eight different banks - as [copy][bin] they would all share the bank of `bin`.
    __shared__ u32 h[4][256 * 8];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 tile = blockIdx.x * 4 + w;
    for (u32 i = lane; i < 8u * 256u; i += 64) h[w][i] = 0;
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
            for (u32 k = 0; k < 16; ++k) if (k < lim) atomicAdd(&h[w][(((v[k >> 2] >> (8u * (k & 3u))) & 255u) << 3) | (lane & 7u)], 1u);
        }
    }
    __syncthreads();
    if (tile < ntiles)
        for (u32 i = lane; i < 256; i += 64) {
            u32 sum = 0;
#pragma unroll
            for (u32 q = 0; q < 8; ++q) sum += h[w][i * 8u + q];
            counts[(u64)tile * 256u + i] = sum;
        }
}

// Exclusive scan of the per-tile symbol counts in symbol-major / tile-minor order, base 1 (cpp:1880-1889).  The counts are
// kept TILE-major, counts[tile][256] (what the count and scatter kernels read and write as whole lines; symbol-major, every
// tile touched 256 lines), so the scan runs down the 256 columns: column sums per block of IBS_TB tiles, symbol bases from
// the column totals, then every block walks its tiles with the sum of the blocks in front of it.
#define IBS_TB 512u
__global__ __launch_bounds__(256) void k_ibwt_scan_partial(const u32* __restrict__ counts, u32 ntiles, u32* __restrict__ bsum /* [blocks][256] */)
{
    const u32 b = blockIdx.x, t = threadIdx.x;
    const u32 t0 = b * IBS_TB, t1 = t0 + IBS_TB < ntiles ? t0 + IBS_TB : ntiles;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    u32 tile = t0;
    for (; tile + 4 <= t1; tile += 4) {
        s0 += counts[(u64)tile * 256u + t]; s1 += counts[(u64)(tile + 1) * 256u + t];
        s2 += counts[(u64)(tile + 2) * 256u + t]; s3 += counts[(u64)(tile + 3) * 256u + t];
    }
    for (; tile < t1; ++tile) s0 += counts[(u64)tile * 256u + t];
    bsum[b * 256u + t] = s0 + s1 + s2 + s3;
}

__global__ __launch_bounds__(256) void k_ibwt_scan_top(const u32* __restrict__ bsum, u32 nblk, u32* __restrict__ symbase /* 256 */)
{
    __shared__ u32 tot[256], base[256];
    const u32 t = threadIdx.x;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    u32 b = 0;
    for (; b + 4 <= nblk; b += 4) { s0 += bsum[b * 256u + t]; s1 += bsum[(b + 1) * 256u + t]; s2 += bsum[(b + 2) * 256u + t]; s3 += bsum[(b + 3) * 256u + t]; }
    for (; b < nblk; ++b) s0 += bsum[b * 256u + t];
    tot[t] = s0 + s1 + s2 + s3;
    __syncthreads();
    scan256_first_wave(tot, base);
    __syncthreads();
    symbase[t] = base[t] + 1u;            // row 0 is the empty suffix (cpp:1891)
}

__global__ __launch_bounds__(256) void k_ibwt_scan_final(u32* __restrict__ counts, u32 ntiles, const u32* __restrict__ bsum, const u32* __restrict__ symbase)
{
    const u32 b = blockIdx.x, t = threadIdx.x;
    const u32 t0 = b * IBS_TB, t1 = t0 + IBS_TB < ntiles ? t0 + IBS_TB : ntiles;
    u32 r0 = symbase[t], r1 = 0, r2 = 0, r3 = 0;
    u32 q = 0;
    for (; q + 4 <= b; q += 4) { r0 += bsum[q * 256u + t]; r1 += bsum[(q + 1) * 256u + t]; r2 += bsum[(q + 2) * 256u + t]; r3 += bsum[(q + 3) * 256u + t]; }
    for (; q < b; ++q) r0 += bsum[q * 256u + t];
    u32 run = r0 + r1 + r2 + r3;
    u32 tile = t0;
    for (; tile + 4 <= t1; tile += 4) {          // four loads in flight, then the dependent adds
        const u32 v0 = counts[(u64)tile * 256u + t], v1 = counts[(u64)(tile + 1) * 256u + t], v2 = counts[(u64)(tile + 2) * 256u + t],
                  v3 = counts[(u64)(tile + 3) * 256u + t];
        counts[(u64)tile * 256u + t] = run; run += v0;
        counts[(u64)(tile + 1) * 256u + t] = run; run += v1;
        counts[(u64)(tile + 2) * 256u + t] = run; run += v2;
        counts[(u64)(tile + 3) * 256u + t] = run; run += v3;
    }
    for (; tile < t1; ++tile) { const u32 v = counts[(u64)tile * 256u + t]; counts[(u64)tile * 256u + t] = run; run += v; }
}

// stable ranked scatter: link[C[c] + rank] = row.  One workgroup per tile of IBWT_WT rows, 2048 rows at a time: rank inside the
// sub-tile from wave ballots + per-wave counters (stable), rows staged in LDS grouped by symbol and written as runs - the
// link table receives whole lines instead of one 4-byte store per row into a few dozen places.
#define IBS_SUB 2048u
__global__ __launch_bounds__(256) void k_ibwt_scatter(const u8* __restrict__ bwt, u32 rows, u32 sent, u32 ntiles,
                                                      const u32* __restrict__ offs, u32* __restrict__ link)
{
    __shared__ u32 cur[256];           // next free link slot of every symbol (this tile)
    __shared__ u32 wcnt[4][256];
    __shared__ u32 tot[256], lstart[256];
    __shared__ u32 stage[IBS_SUB];
    __shared__ u8 sbin[IBS_SUB];
    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63u;
    const u32 tile = blockIdx.x;
    if (tile == 0 && t == 0) link[0] = sent;      // cpp:1891
    cur[t] = offs[(u64)tile * 256u + t];
    const u32 beg = tile * IBWT_WT;
    const u32 end = (rows - beg < IBWT_WT) ? rows : beg + IBWT_WT;
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (u32 sb = beg; sb < end; sb += IBS_SUB) {
        __syncthreads();
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) wcnt[w2][t] = 0;
        __syncthreads();
        u32 cs[8], pos[8];
        u32 vmask = 0;
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            const u32 r = sb + w * 512u + k * 64u + lane;
            const bool valid = r < end && r != sent;
            cs[k] = valid ? ibwt_sym(bwt, r, sent) : 0u;
            vmask |= (u32)valid << k;
        }
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            const bool valid = (vmask >> k) & 1u;
            const u32 c = cs[k];
            u64 mask = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (c >> b) & 1u;
                const u64 bal = __ballot(bit);
                mask &= bit ? bal : ~bal;
            }
            pos[k] = 0;
            if (valid) {
                const int leader = __ffsll((long long)mask) - 1;
                u32 old = 0;
                if ((int)lane == leader) old = atomicAdd(&wcnt[w][c], (u32)__popcll(mask));
                old = __shfl(old, leader, 64);
                pos[k] = old + (u32)__popcll(mask & lt_mask);
            }
        }
        __syncthreads();
        {
            u32 o = 0;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) { const u32 v = wcnt[w2][t]; wcnt[w2][t] = o; o += v; }
            tot[t] = o;
        }
        __syncthreads();
        scan256_first_wave(tot, lstart);
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < 8; ++k)
            if ((vmask >> k) & 1u) {
                const u32 c = cs[k], slot = lstart[c] + wcnt[w][c] + pos[k];
                stage[slot] = sb + w * 512u + k * 64u + lane;
                sbin[slot] = (u8)c;
            }
        __syncthreads();
        const u32 total = lstart[255] + tot[255];
        for (u32 s = t; s < total; s += 256u) { const u32 c = sbin[s]; link[cur[c] + (s - lstart[c])] = stage[s]; }
        __syncthreads();
        cur[t] += tot[t];
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
// thread, cpp:1922,1976-2018).  ONE since round 3: the walk is bound by dependent random sector reads, 2048 lanes per CU
// with one chain each already reach what the chip gives (tools/microbench/exp_random_lines.hip), and the registers of a
// second chain are better spent on 64-byte pieces.
#ifndef IBWT_CW
#define IBWT_CW 1024u
#endif
#ifndef IBWT_NCH
#define IBWT_NCH 1                  // chains per lane (2048 lanes per CU walk one chain each: more adds nothing, see profiles/HISTORY.md, tried and rejected)
#endif
#ifndef IBWT_PIECE
#define IBWT_PIECE 64u              // bytes per scattered store of a chain (a power of two, 16 .. 128): one whole 64-byte sector
#endif
__global__ __launch_bounds__(256) void k_ibwt_walk(const u32* __restrict__ link, const u32* __restrict__ offs /* [ntiles][256]: C[c] = offs[c] (tile 0) */,
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
    s_C[t] = offs[t];
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
    constexpr u32 PW = IBWT_PIECE / 8u;          // 8-byte words per piece
    u64 acc[IBWT_NCH][PW];             // the last (len % IBWT_PIECE) bytes met: bytes leave as aligned pieces - one whole 64-byte
                                       // sector per 64 hops of a chain (the walk is bound by random DRAM accesses and every
                                       // scattered store is one more: 8-byte stores 31 ms, 32-byte pieces 29.4, 64-byte 28.1)
#pragma unroll
    for (int c = 0; c < IBWT_NCH; ++c) {
        id[c] = IBWT_PULL(); my[c] = id[c]; len[c] = 0; cur[c] = id[c] < K ? ibwt_start(id[c], sent, kreg) : 0u;
#pragma unroll
        for (u32 w = 0; w < PW; ++w) acc[c][w] = 0;
    }
#if defined(IBWT_EXP) && (IBWT_EXP & 1)      // experiment: no byte stores
#define IBWT_FLUSH(c, at) do { if (acc[c][0] == 0x123456789abcull) segbuf[my[c]] = 1; _Pragma("unroll") for (u32 w_ = 0; w_ < PW; ++w_) acc[c][w_] = 0; } while (0)
#else
#if defined(IBWT_NT) && (IBWT_NT & 1)        // experiment: the pieces leave as non-temporal stores
#define IBWT_STORE(p_, v_) __builtin_nontemporal_store(v_, p_)
#else
#define IBWT_STORE(p_, v_) (*(p_) = (v_))
#endif
#define IBWT_FLUSH(c, at) do { u64* o_ = reinterpret_cast<u64*>(segbuf + (u64)my[c] * IBWT_CW + (at)); \
        typedef u64 v2u64 __attribute__((ext_vector_type(2))); \
        _Pragma("unroll") for (u32 w_ = 0; w_ < PW; w_ += 2) { \
            v2u64 q_; q_.x = acc[c][w_]; q_.y = acc[c][w_ + 1]; \
            IBWT_STORE(reinterpret_cast<v2u64*>(o_ + w_), q_); \
            acc[c][w_] = acc[c][w_ + 1] = 0; } } while (0)
#endif
    for (;;) {
        bool any = false;
        u32 nx[IBWT_NCH];
#pragma unroll
#if defined(IBWT_NT) && (IBWT_NT & 2)        // experiment: every link is read exactly once - non-temporal loads
        for (int c = 0; c < IBWT_NCH; ++c) { nx[c] = 0; if (id[c] < K) { nx[c] = __builtin_nontemporal_load(link + cur[c]); any = true; } }
#else
        for (int c = 0; c < IBWT_NCH; ++c) { nx[c] = 0; if (id[c] < K) { nx[c] = link[cur[c]]; any = true; } }     // the dependent loads, all in flight together
#endif
        if (!any) break;
#pragma unroll
        for (int c = 0; c < IBWT_NCH; ++c) {
            if (id[c] >= K) continue;
            // symbol of the row I am leaving (independent of the load above)
#if defined(IBWT_EXP) && (IBWT_EXP & 2)      // experiment: no symbol search
            u32 sy = cur[c] & 255u;
#else
            u32 sy = s_T[cur[c] >> shift];
            while (cur[c] >= s_C[sy + 1]) ++sy;
#endif
            {
                const u64 v = (u64)sy << (8u * (len[c] & 7u));
                const u32 wsel = (len[c] >> 3) & (PW - 1u);
#pragma unroll
                for (u32 w = 0; w < PW; ++w) acc[c][w] |= wsel == w ? v : 0ull;
            }
            ++len[c];
            if ((len[c] & (IBWT_PIECE - 1u)) == 0) IBWT_FLUSH(c, len[c] - IBWT_PIECE);
            const u32 r = nx[c];
            if (ibwt_marked(r, sent)) {
                if (len[c] & (IBWT_PIECE - 1u)) IBWT_FLUSH(c, len[c] & ~(IBWT_PIECE - 1u));      // (tail bytes beyond len are never read; the buffer has room: len < IBWT_CW here)
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
#undef IBWT_STORE
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

