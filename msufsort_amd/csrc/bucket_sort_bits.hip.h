// k_sort_bits: LDS sort of one two-byte bucket whose keys are spread out (random-like input) - the counterpart of
// multikey_quicksort for one partition (reference msufsort.cpp:488-642) and the hot kernel of the headline.  Included by
// sa_kernels.hip.h.
//
// Why another kernel.  k_sort_fast2 (below in sa_kernels.hip.h) ranks every record with a byte-packed counting table and
// five straight-line probes: 112 vector + 43 scalar + 14 LDS instructions per record (rocprofv3 SQ counters, round 2).
// Neither HBM (the kernel runs as long with its loads and stores removed), nor the LDS array (45 % busy), nor the vector
// ALUs limit it: a wave issues one instruction every 4-5 clocks, a CU holds 16 waves of this shape, and 2,700 instructions per
// wave and segment separated by ten barriers leave every pipe waiting for the instruction streams.  So the design goal is
// INSTRUCTIONS PER RECORD, and dependent LDS round trips per wave:
//
//   bitmap rank.  The kbits key bits that can still differ are cut down to 18 (16 for the small shape): 2^18 buckets for at
//   most 18,432 records, one BIT each.  A 32-bit LDS word describes 16 neighbouring buckets.
//     A  every record ADDS  (1 << bucket & 15) | 1 << 20  to its word - one non-returning LDS atomic, nothing to wait for:
//        bits 0..15 collect the buckets, bits 20..31 count the arrivals, bits 16..19 catch the carries of colliding adds
//        (a second arrival in a bucket flips bits upwards instead of setting one: popcount(bits 0..19) < arrivals).
//     S  one scan over the words (16-byte LDS accesses): arrivals -> exclusive prefix = the word's first row; a word whose
//        popcount disagrees with its arrivals is DIRTY.  The word becomes [31] dirty | [30:16] first row | [15:0] bits.
//     B  a record of a clean word knows its FINAL row from ONE LDS word: first row + popcount(bits below its own); it stores its
//        suffix index there.  Records of dirty words (8 % on uniform keys: two records of one bucket, or a neighbour of such
//        a pair) store to a scratch row and are pushed to a list {key, index}: one reservation per wave and segment.
//     D  the list is processed densely, one entry per lane: the entries of a word claim the slots of the word's row range
//        (the word's bit field, zeroed by the scan, is the claim counter), post their keys there, and rank themselves by
//        counting smaller keys among the posted ones; equal keys are tie runs and go to the next round like everywhere else.
//   Rows leave coalesced, as aligned 8-byte stores.  The next segment's records are loaded while this one is sorted.
//
// LDS per workgroup (1024 threads, 16,384 words, segments up to 18,432 records): words 64 KiB, rows 72 KiB, dirty list
// 20 KiB, tie list 1.5 KiB = 158 KiB: one workgroup per CU.  Static LDS: every address is a compile-time constant.
//
// Handed back to k_sort_mid through fb_list (as k_sort_fast2 does): segments longer than LEN_MAX, more than 30 varying key
// bits, a dirty list or tie list that overflows (skewed keys), runs of more than 255 equal keys.
#pragma once

// class C: 1024 threads x 18 records = every class-C segment (18,432), 2560 dirty-list entries (a full segment: 2030 +- 80)
#define BITS_C_SHAPE 1024, 18, 18432, 2560, 128
// class B: 256 threads x 18 records = every class-B segment (4608), 640 dirty-list entries; 40,912 B of LDS: four workgroups per CU
#define BITS_B_SHAPE 256, 18, 4608, 640, 32

// Diagnostic build (-DBITS_PROF): clock64() per phase, summed over the workgroups' first threads, printed by the engine.
#ifdef BITS_PROF
__device__ unsigned long long g_bits_prof[16];
#define BPROF(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); prof_acc[i] += now_ - prof_last; prof_last = now_; } } while (0)
#else
#define BPROF(i) do { } while (0)
#endif

template <int THREADS, int ITEMS, int LEN_MAX, int LCAP, int TL>
__global__ __launch_bounds__(THREADS, 1024 / THREADS) void k_sort_bits(RecBufs bufs, const Desc* __restrict__ list, u32 nseg,
                                                       u32* __restrict__ sa_out, Emit em, u32* __restrict__ counters,
                                                       u32* __restrict__ fb_list, u32 fb_cnt_idx)
{
    constexpr int W = THREADS / 64;
    constexpr int NW = THREADS * 16;                  // LDS words, 16 buckets each; a thread scans 4 quads of words
    constexpr int NL = ITEMS / 2;
    constexpr int LPT = (LCAP + THREADS - 1) / THREADS;      // dirty-list entries per thread
    constexpr int BATCH = 6;                          // LDS reads in flight per lane in phase B
    static_assert(ITEMS % 2 == 0 && ITEMS % BATCH == 0 && (THREADS & (THREADS - 1)) == 0 && W * 4 <= 64, "shapes");
    static_assert(LEN_MAX <= THREADS * ITEMS && LEN_MAX + 64 < 32768, "segment length limit (rows are 15-bit fields)");
    constexpr u32 GB = THREADS == 1024 ? 18u : THREADS == 512 ? 17u : THREADS == 256 ? 16u : THREADS == 128 ? 15u : 14u;      // log2(buckets)
    static_assert((1u << GB) == (u32)NW * 16u, "GB");
    constexpr u32 TRASH_ROW = LEN_MAX + 32;           // where lanes outside the segment and records of dirty words store
    struct Lds {
        u32 bw[NW + 4];               // the words, at LDS address 0 (a word's address is a shifted key, nothing to add);
                                      // [NW] = len << 16 closes the last word's row range; [NW + 1]: word of the lanes outside the segment
        u32 out[LEN_MAX + 64];        // rows (suffix indices in final order); mailbox of the dirty words
        uint2 lst[LCAP];              // {key, index} of the records of dirty words
        u32 tl[3 * TL];               // tie list {index, rs | rl << 16 | ro << 24, local offset}
        u32 tot[64];                  // wave totals of the scan: [quad row][wave]
        u32 misc[16];
    };
    __shared__ __attribute__((aligned(16))) Lds L;
    u32* const bw = L.bw; u32* const out = L.out; uint2* const lst = L.lst; u32* const tl = L.tl; u32* const tot = L.tot; u32* const misc = L.misc;

    u32 t = threadIdx.x;
#define BITS_P(j) (((((u32)(j) >> 1) * (u32)THREADS + t) << 1) + ((u32)(j) & 1u))
#define BITS_SRC(dd) (reinterpret_cast<const unsigned char*>((((dd).buf & 3u) == 0u ? bufs.p[0] : ((dd).buf & 3u) == 1u ? bufs.p[1] : bufs.p[2]) + (dd).rec_off))
    u32 seg = blockIdx.x;
    if (seg >= nseg) return;
    Desc d = list[seg];
    Desc dn = list[seg + gridDim.x < nseg ? seg + gridDim.x : seg];      // unconditional, clamped: stays a scalar load
    u64 nrec[ITEMS];
    {
        const unsigned char* src = BITS_SRC(d);
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const u32 p = BITS_P(2 * q);
            const Rec2 v = *reinterpret_cast<const Rec2*>(src + (p < d.len ? p * 8u : 0u));    // (may read one record past the segment)
            nrec[2 * q] = v.a; nrec[2 * q + 1] = v.b;
        }
    }
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, in a scalar register
    // The words start out zero and every segment leaves them zero again: they are cleared while its rows go out (they are dead from
    // the end of phase D on), under the barrier that closes the segment - not in a phase and behind a barrier of their own at the
    // start of the next one (round 6: 1,900 of a segment's 25,700 clocks, profiles/r06_bits_prof.txt)
#define BITS_CLEAR() do { uint4* z_ = reinterpret_cast<uint4*>(bw); const uint4 z4_ = {0u, 0u, 0u, 0u}; \
        _Pragma("unroll") for (u32 i_ = 0; i_ < 4; ++i_) z_[i_ * THREADS + threadIdx.x] = z4_; if (threadIdx.x == 0) z_[NW / 4] = z4_; } while (0)
    BITS_CLEAR();
    __syncthreads();
#ifdef BITS_PROF
    __shared__ unsigned long long prof_acc[16];
    unsigned long long prof_last = clock64();
    if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) prof_acc[i] = 0;
#endif
    for (;;) {
        asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(threadIdx.x));      // (address arithmetic redone per segment, not hoisted into registers)
        const u32 lane = t & 63u;
        const u32 len = d.len, sa_off = d.sa_off, cur = seg, kbits = (d.buf >> 8) & 255u;
        const u32 kmask = kbits >= 32 ? 0xffffffffu : ((1u << kbits) - 1u);
        const u32 sh = kbits > GB ? kbits - GB : 0u;
        const u32 key_out = (((u32)NW + 1u) * 16u) << sh;        // key of a lane outside the segment: its bucket lies in word NW + 1
        u32 key[ITEMS], idx[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { key[j] = (u32)(nrec[j] >> 32) & kmask; idx[j] = (u32)nrec[j]; }
        seg += gridDim.x;
        const bool more = seg < nseg;
        d = dn;
        // the next segment's records travel while this one is sorted, a few loads behind each phase (a CU accepts
        // vector-memory instructions slowly: eighteen back to back stall every wave)
        const unsigned char* nsrc = BITS_SRC(d);
        const u32 nlen = more ? d.len : 0u;
#if defined(BITS_EXP) && (BITS_EXP & 1)      // experiment: no record loads (keys from a hash): what does the compute alone cost?
#define BITS_LOAD(from, upto) do { _Pragma("unroll") for (int q_ = (from); q_ < (upto) && q_ < NL; ++q_) { const u32 p_ = BITS_P(2 * q_); u32 h_ = (p_ + seg * 18432u) * 0x9E3779B1u; h_ ^= h_ >> 15; h_ *= 0x85EBCA77u; h_ ^= h_ >> 13; nrec[2 * q_] = ((u64)(h_ >> 8) << 32) | p_; h_ *= 0xC2B2AE3Du; h_ ^= h_ >> 16; nrec[2 * q_ + 1] = ((u64)(h_ >> 8) << 32) | (p_ + 1u); } } while (0)
#else
#define BITS_LOAD(from, upto) do { _Pragma("unroll") for (int q_ = (from); q_ < (upto) && q_ < NL; ++q_) { const u32 p_ = BITS_P(2 * q_); const Rec2 v_ = *reinterpret_cast<const Rec2*>(nsrc + (p_ < nlen ? p_ * 8u : 0u)); nrec[2 * q_] = v_.a; nrec[2 * q_ + 1] = v_.b; } } while (0)
#endif
        constexpr int LB = (NL + 4) / 5 > 1 ? (NL + 4) / 5 : 2;      // pair loads per batch, five batches
        dn = list[seg + gridDim.x < nseg ? seg + gridDim.x : nseg - 1u];
        const u32 nrows = (len + 127u) >> 7;                     // a wave covers 128 consecutive positions per pair load
        const int rows = 2 * (nrows > wv ? (int)((nrows - wv + W - 1) / W) : 0);     // items of this wave that can be inside the segment (scalar)
        bool ok = kbits <= 30u && len != 0 && len <= (u32)LEN_MAX;       // (key_out must fit 32 bits: (2^GB + 16) << (kbits - GB))
        u32 res_t = 0, res_s = 0;

        if (ok) {
            // (the flags: nobody reads them between the barrier that closed the previous segment and barrier (2); the first writer
            // is the scan's last thread, behind (3))
            if (t < 16) misc[t] = 0;
            // the lanes of the segment's last, partly filled row that lie outside it get the key of the spare word: from
            // here on no phase needs a bounds test
            if (len & 127u) {
#pragma unroll
                for (int q = 0; q < NL; ++q)
                    if ((u32)q * W + wv == nrows - 1u) {
                        if (BITS_P(2 * q) >= len) key[2 * q] = key_out;
                        if (BITS_P(2 * q + 1) >= len) key[2 * q + 1] = key_out;
                    }
            }
        }
        const bool touched = ok;                                 // phase A runs: the words need clearing afterwards
        BPROF(0);
        BITS_LOAD(0, LB);
        const u32 shw = sh + 2u;                                 // byte offset of a key's word: (key >> shw) & ~3
#define BITS_WORD(k) (*reinterpret_cast<u32*>(reinterpret_cast<unsigned char*>(bw) + (((k) >> shw) & 0x3fffcu)))
        if (ok) {                                                // ---- A: one add per record, nothing returned
#pragma unroll
            for (int q = 0; q < NL; ++q)
                if (2 * q < rows) {
#pragma unroll
                    for (int j = 2 * q; j < 2 * q + 2; ++j)
                        atomicAdd(&BITS_WORD(key[j]), (1u << ((key[j] >> sh) & 15u)) | 0x100000u);
                }
            __syncthreads();                                                        // (2)
        }
        BPROF(1);
        BITS_LOAD(LB, 2 * LB);
        if (ok) {                                                // ---- S: first row of every word
            // thread t owns the word quads (k THREADS + t), k = 0..3: 16-byte LDS accesses, consecutive lanes on consecutive quads
            uint4 q4[4];
            u32 s4[4], inc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q4[k] = reinterpret_cast<const uint4*>(bw)[k * THREADS + t];
                s4[k] = (q4[k].x >> 20) + (q4[k].y >> 20) + (q4[k].z >> 20) + (q4[k].w >> 20);
                inc[k] = wave_incl_scan_dpp(s4[k]);
            }
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 4; ++k) tot[k * W + wv] = inc[k];
            }
            __syncthreads();                                                        // (3)
            // every wave scans the 4 W totals itself (one DPP scan) and picks its own bases out of the result
            const u32 tv = lane < 4u * W ? tot[lane] : 0u;
            const u32 texc = wave_incl_scan_dpp(tv) - tv;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32 x = (u32)__builtin_amdgcn_readlane((int)texc, k * W + wv) + inc[k] - s4[k];
                u32 w[4] = {q4[k].x, q4[k].y, q4[k].z, q4[k].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 cnt = w[i] >> 20;
                    const bool dirty = (u32)__popc(w[i] & 0xfffffu) != cnt;
                    w[i] = dirty ? ((x | 0x8000u) << 16) : ((w[i] & 0xffffu) | (x << 16));      // (dirty: bit field zeroed, it becomes the claim counter)
                    x += cnt;
                }
                uint4 r; r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3];
                reinterpret_cast<uint4*>(bw)[k * THREADS + t] = r;
                if (k == 3 && t == THREADS - 1) {
                    bw[NW] = x << 16; bw[NW + 1] = TRASH_ROW << 16;
                    // x = len, unless a word took 4096 arrivals or more and its 12-bit count wrapped (all keys alike): k_sort_mid's case
                    if (x != len) misc[1] = 1u;
                }
            }
            __syncthreads();                                                        // (4)
        }
        BPROF(2);
        BITS_LOAD(2 * LB, 3 * LB);
        u32 nl = 0;
        if (ok && misc[1]) ok = false;
        if (ok) {                                                // ---- B: final rows of the clean words' records
            u32 dmask = 0;                                       // items of this thread that sit in dirty words
#pragma unroll
            for (int j0 = 0; j0 < ITEMS; j0 += BATCH)
                if (j0 < rows) {
                    u32 e[BATCH];
#pragma unroll
                    for (int b = 0; b < BATCH; b += 2)
                        if (j0 + b < rows) { e[b] = BITS_WORD(key[j0 + b]); e[b + 1] = BITS_WORD(key[j0 + b + 1]); }
#pragma unroll
                    for (int b = 0; b < BATCH; b += 2)
                        if (j0 + b < rows) {                     // (scalar test per pair: the registers of rows behind the segment hold anything)
#pragma unroll
                            for (int c = b; c < b + 2; ++c) {
                                const int j = j0 + c;
                                const u32 g = key[j] >> sh;
                                const u32 row = (u32)__popc(e[c] & ((1u << (g & 15u)) - 1u)) + ((e[c] >> 16) & 0x7fffu);
                                out[(int)e[c] < 0 ? TRASH_ROW : row] = idx[j];
                                dmask |= (e[c] >> 31) << j;
                            }
                        }
                }
            // ONE list reservation per wave and segment: wave scan of the per-thread counts
            const u32 dc = (u32)__popc(dmask);
            const u32 dinc = wave_incl_scan_dpp(dc);
            const u32 wtot = (u32)__builtin_amdgcn_readlane((int)dinc, 63);
            if (wtot) {                                          // (wave-uniform)
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&misc[0], wtot);
                const u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)base);
                // (its own flag: misc[1] is READ right behind barrier (4) - a wave that is still on its way to that read must not see
                // what a faster wave sets here, or the two take different sides of `ok` and of every barrier that follows.  Round 4:
                // that race existed since round 3 and showed on inputs that are one text twice at sizes where the 4608-record shape
                // takes 17-bit children: waves one barrier apart, phase-A adds of the next segment landing in words phase D was
                // claiming, a 2^32-iteration loop per segment - 6 minutes for 384 MiB)
                if (wbase + wtot > (u32)LCAP) { if (lane == 0) misc[10] = 1u; }
                else {
                    u32 pos = wbase + dinc - dc;
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j)
                        if ((dmask >> j) & 1u) { uint2 r; r.x = key[j]; r.y = idx[j]; lst[pos] = r; ++pos; }
                }
            }
            __syncthreads();                                                        // (5)
            nl = __builtin_amdgcn_readfirstlane(misc[0]);      // (uniform: list passes nobody takes part in are branched over)
#ifdef BITS_PROF
            if (threadIdx.x == 0) { prof_acc[10] += nl; prof_acc[12] += misc[1]; prof_acc[14] += nl > 1700u; prof_acc[15] += nl > 2048u; if (nl > 2048u && prof_acc[13] == 0) prof_acc[13] = ((unsigned long long)len << 32) | nl; }
#endif
            if (misc[10]) ok = false;                            // skewed keys: k_sort_mid takes the segment (written before barrier (5) only)
        }
        BPROF(3);
        BITS_LOAD(3 * LB, 4 * LB);
        if (ok && nl) {                                          // ---- D: the records of the dirty words, one list entry per lane
            u32 dk[LPT], di[LPT], db[LPT], dc[LPT], ds[LPT];
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const u32 e = t + (u32)i * THREADS;
                dk[i] = 0; di[i] = 0; db[i] = TRASH_ROW; dc[i] = 0; ds[i] = 0;
                if ((u32)i * THREADS >= nl) continue;
                if (e < nl) {
                    const uint2 r = lst[e];
                    dk[i] = r.x; di[i] = r.y;
                    u32* const dwp = &BITS_WORD(r.x);
                    const u32 w0 = atomicAdd(dwp, 1u), w1 = dwp[1];                // claim a slot of the word's row range
                    db[i] = (w0 >> 16) & 0x7fffu;
                    dc[i] = ((w1 >> 16) & 0x7fffu) - db[i];
                    ds[i] = w0 & 0xffffu;
                    if (dc[i] > (u32)LEN_MAX || ds[i] >= dc[i]) { misc[9] = 1u; dc[i] = 0; ds[i] = 0; db[i] = TRASH_ROW; }      // (belt and braces: a word that is not
                                                                                                                  // what the scan left must never be looped on)
                    out[db[i] + ds[i]] = dk[i];                  // the word's rows as mailbox
                }
            }
            __syncthreads();                                                        // (6)
            BPROF(4);
            u32 dl[LPT], dq[LPT];                                // rank among the word's records; equal keys: run length | offset << 16
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                dl[i] = 0; dq[i] = 0;
                if ((u32)i * THREADS >= nl) continue;
                const bool in = t + (u32)i * THREADS < nl;
                // the first eight posted keys at once (a dirty word holds 3 records on average); what lies behind the word's
                // rows is read too and masked
                u32 c[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) c[q] = out[db[i] + q];
                u32 lt = 0, eq = 0, ro = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool v = (u32)q < dc[i];
                    lt += v & (c[q] < dk[i]); eq += v & (c[q] == dk[i]); ro += v & (c[q] == dk[i]) & ((u32)q < ds[i]);
                }
                if (__ballot(in && dc[i] > 8u)) {                // (rare) longer words
#pragma nounroll
                    for (u32 q = 8; q < dc[i]; ++q) {
                        const u32 cc = out[db[i] + q];
                        lt += cc < dk[i]; eq += cc == dk[i]; ro += (cc == dk[i]) & (q < ds[i]);
                    }
                }
                if (in) {
                    dl[i] = lt; dq[i] = eq | (ro << 16);
                    if (eq > 255u) misc[1] = 1u;                 // (a run the tie list cannot describe)
                }
            }
            __syncthreads();                                                        // (7)
            BPROF(6);
#pragma unroll
            for (int i = 0; i < LPT; ++i)
                if (t + (u32)i * THREADS < nl) {
                    const u32 eq = dq[i] & 0xffffu, ro = dq[i] >> 16;
                    out[db[i] + dl[i] + ro] = di[i];
                    if (eq > 1u) {
                        const u32 slot = atomicAdd(&misc[4], 1u);
                        u32 loc = 0;
                        if (ro == 0) loc = atomicAdd(&misc[eq <= TINY_MAX ? 2 : 3], eq);
                        if (slot < (u32)TL) { tl[3 * slot] = di[i]; tl[3 * slot + 1] = (db[i] + dl[i]) | ((eq & 255u) << 16) | (ro << 24); tl[3 * slot + 2] = loc; }
                    }
                }
            __syncthreads();                                                        // (8)
#ifdef BITS_PROF
            if (threadIdx.x == 0) { prof_acc[11] += misc[4]; }
#endif
            if (misc[1] || misc[9] || misc[4] > (u32)TL) ok = false;
        }
        BPROF(7);
        BITS_LOAD(4 * LB, NL);
        if (touched) BITS_CLEAR();                               // (block-uniform; nothing reads the words any more)
        if (ok) {                                                // ---- rows out
            const u32 nt = misc[4];
            if (t == 0 && nt != 0) {
                u32 bad = 0;
                if (misc[2]) res_t = atomicAdd(&counters[em.pool_cnt_idx], misc[2]);
                if (misc[3]) res_s = atomicAdd(&counters[em.seg_cnt_idx], misc[3]);
                if ((u64)res_t + misc[2] > em.pool_cap) { atomicOr(&counters[C_ERR], 32u); bad = 1; }
                if ((u64)res_s + misc[3] > em.seg_cap) { atomicOr(&counters[C_ERR], 64u); bad = 1; }
                misc[6] = res_t; misc[7] = res_s; misc[8] = bad;
            }
            {   // 8 bytes per lane at 8-byte aligned addresses: lane pairs start at an even ROW of the array
                const u32 shift = sa_off & 1u;
                u32* outp = sa_out + sa_off;
                if (shift && t == 0) outp[0] = out[0];
#pragma unroll
                for (int q = 0; q < NL; ++q)
                    if (2 * q < rows) {
                        const u32 p = BITS_P(2 * q) + shift;
#if defined(BITS_EXP) && (BITS_EXP & 2)      // experiment: no row stores
                        if (p + 1 < len) { if ((out[p] ^ out[p + 1]) == 0x12345u) outp[p] = 1; }
#else
                        if (p + 1 < len) {
                            uint2 v; v.x = out[p]; v.y = out[p + 1];
                            *reinterpret_cast<uint2*>(outp + p) = v;
                        } else if (p < len) outp[p] = out[p];
#endif
                    }
            }
            if (nt) {                                            // block-uniform: tie runs go to the next round
                __syncthreads();                                 // rows are out: out[run start] now carries the run's local offset
                for (u32 i = t; i < nt; i += THREADS) { const u32 w1 = tl[3 * i + 1]; if ((w1 >> 24) == 0) out[w1 & 0xffffu] = tl[3 * i + 2]; }
                __syncthreads();
                if (!misc[8]) {
                    const u32 base_t = misc[6], base_s = misc[7];
                    for (u32 i = t; i < nt; i += THREADS) {
                        const u32 id = tl[3 * i], w1 = tl[3 * i + 1];
                        const u32 rs = w1 & 0xffffu, rl = (w1 >> 16) & 255u, ro = w1 >> 24;
                        if (rl <= TINY_MAX) {
                            const u32 o = base_t + out[rs] + ro;
                            em.pool_rec[o] = (u64)id;
                            em.pool_hdr[o] = pack_hdr(sa_off + rs, rl, ro);
                        } else {
                            const u32 o = base_s + out[rs] + ro;
                            em.seg_rec[o] = (u64)id;
                            if (ro == 0) { const Desc nd = {o, rl, sa_off + rs, em.seg_buf}; push_desc(em.lists, counters, class_of(rl), nd); }
                        }
                    }
                }
            }
        }
        BPROF(8);
        if (!ok && len != 0 && t == 0) fb_list[atomicAdd(&counters[fb_cnt_idx], 1u)] = cur;   // leave it to k_sort_mid
        if (!more) break;
        __syncthreads();            // everyone is done with this segment's LDS before it is reset
        BPROF(9);
    }
#ifdef BITS_PROF
    if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_bits_prof[i], prof_acc[i]);
#endif
#undef BITS_P
#undef BITS_SRC
#undef BITS_LOAD
#undef BITS_WORD
#undef BITS_CLEAR
}
