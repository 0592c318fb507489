/*
 * msufsort_hip.h - C-ABI of the MI355X (gfx950) suffix-array / BWT engine.
 *
 * This is the drop-in boundary for the hot path of michaelmaniscalco/msufsort.  The
 * reference has no FFI layer; its public surface is the C++ class maniscalco::msufsort
 * (reference src/library/msufsort/msufsort.h:42-75) and three free templates
 * (msufsort.h:403-476).  include/library/msufsort.h in this repository keeps that C++
 * surface verbatim and forwards to the entry points below; any other host language binds
 * the same symbols (see INTEGRATION.md for the stubs).
 *
 * Conventions (identical to the reference):
 *   - SA has n+1 int32 entries, SA[0] = n (the empty suffix), SA[1..n] = suffixes in
 *     lexicographic order of unsigned bytes, a proper prefix sorting first.
 *   - BWT is n bytes with the sentinel row removed; the sentinel row index (the row r
 *     with SA[r] == 0, always in [1, n]) is returned separately.
 *   - LCP[i] = lcp(suffix SA[i+1], suffix SA[i+2]) for i in [0, n-2]; LCP[n-1] = 0.
 *
 * All functions return 0 on success or a negative msufsort_hip_status; nothing throws
 * across this boundary.  `_dev` variants take device pointers (HBM-resident data);
 * the others take host pointers (pageable is fine) and stage through HBM.
 * A context owns one HIP stream and a workspace; calls on one context are serialised,
 * distinct contexts are independent.
 */
#ifndef MSUFSORT_HIP_H
#define MSUFSORT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum msufsort_hip_status {
    MSUFSORT_HIP_OK = 0,
    MSUFSORT_HIP_ERR_NO_DEVICE = -1,     /* no gfx950 device visible: the product path has no CPU fallback */
    MSUFSORT_HIP_ERR_BAD_ARG = -2,
    MSUFSORT_HIP_ERR_TOO_LARGE = -3,     /* n > 2^31 - 2 for the int32 entry points, n > 2^40 - 2 for the int64 ones */
    MSUFSORT_HIP_ERR_HIP = -4,           /* a HIP runtime call failed; see msufsort_hip_last_error */
    MSUFSORT_HIP_ERR_NOMEM = -5,
    MSUFSORT_HIP_ERR_INTERNAL = -6,      /* internal capacity/consistency check tripped */
    MSUFSORT_HIP_ERR_UNSUPPORTED = -7    /* e.g. deep ties in a sharded (n_shards > 1) build */
} msufsort_hip_status;

typedef struct msufsort_hip_ctx msufsort_hip_ctx;

/* Tunables; zero-initialise and override what you need. */
typedef struct msufsort_hip_opts {
    int32_t device;            /* HIP device ordinal (default 0) */
    int32_t shard;             /* msufsort_hip_make_sa_shard*_dev: this call builds shard `shard` of `n_shards` (16-bit-key range
                                  split).  msufsort_hip_make_sa_i32_dev: shard = -1 builds ALL n_shards logical shards one after
                                  the other on the context's GPU (bounded workspace; distributed prefix doubling) */
    int32_t n_shards;          /*   0 or 1 = whole array; msufsort_hip_make_sa_i64_dev: at least this many logical shards */
    int32_t text_rounds;       /* key-gather rounds before switching to prefix doubling; 0 = adaptive (switch when the
                                  tied set stops shrinking past the depth the alphabet needs, at the latest after 24) */
    int32_t verbose;           /* 1: per-round statistics on stderr */
    int32_t force_wide;        /* int64 entry points: use the wide (40-bit index) engine also below 2^31 - 1 bytes (parity tests) */
    int32_t two_stage;         /* msufsort_hip_make_sa_i32_dev (and what is built on it: forward BWT, host-pointer entry points):
                                  0 = sort only the B* suffixes and induce the others (the reference's two stages, cpp:1496-1555 +
                                  cpp:646-1057) when the input looks like text or DNA (4..128 byte values in use; at least 136 MiB,
                                  320 MiB below 16 values; no very long runs of one byte), 1 = whenever possible, -1 = never; inputs that do not suit (or
                                  whose B* suffixes tie too deep) are sorted completely, as before */
    int32_t reuse_plan;        /* msufsort_hip_make_sa_shard*_dev: 1 = this call builds ANOTHER shard of the text the previous shard call on this
                                  context planned (same buffer, same n, same n_shards, contents unchanged - the caller's promise): the 16-bit
                                  histogram and the plan of the cuts are taken from that call instead of a new pass over the text.  A rank
                                  that sorts its key range as several sub-shards so that finished sub-slices travel while the next one is
                                  sorted (msufsort_amd/dist.py, sub_shards) sets it from its second sub-shard on; ignored (a full plan runs)
                                  when the context holds no matching plan */
    int32_t reserved[8];
} msufsort_hip_opts;

/* Per-phase device time of the last build on this context (hipEvent, milliseconds), plus
 * the algorithmic byte counts bench.py uses for the roofline. */
typedef struct msufsort_hip_timings {
    double total_ms;
    double hist16_ms;          /* 16-bit radix histogram: reads n bytes */
    double scatter0_ms;        /* first-byte scatter: reads n, writes 8*m */
    double scatter1_ms;        /* second-byte scatter: reads 8*m, writes 8*m */
    double bucket_sort_ms;     /* round-0 LDS sorts of the 16-bit buckets */
    double refine_ms;          /* all later rounds (key gathers / prefix doubling) */
    double other_ms;           /* single-process sharded builds: wall time of the distributed doubling phase */
    int64_t n;
    int64_t m;                 /* suffixes sorted by the radix path (n minus trailing 0x00 run) */
    int32_t rounds;            /* rounds after round 0 */
    int32_t doubling_rounds;
    int64_t unresolved_after_round0;
    int64_t stop_depth;        /* depth (bytes every remaining tie group shares) at which a sharded build stopped its key rounds */
    int64_t logical_shards;    /* logical shards of the last build */
    int64_t gathered_records;  /* records whose key was gathered, summed over the rounds after round 0 (each: 4 B index read,
                                  one 64 B sector of text or ranks, 8 B record written, 4 B row written by the sorts) */
    int64_t ibwt_walk_us;      /* inverse BWT: microseconds of the chain walk (k_ibwt_walk) ... */
    int64_t ibwt_total_us;     /* ... and of the whole inverse */
    /* two-stage builds (B* sort + induction; the sort phases above then cover the B* suffixes only, other_ms is the induction) */
    int64_t bstar_suffixes;    /* 0: the last build sorted all suffixes */
    int64_t induction_launches;/* level launches of the two passes */
    int64_t b_suffixes;        /* B-type suffixes (rows pass B reads) */
    double front_ms;           /* two-stage builds: typing + histograms before stage 1 (k_hist16, k_types, k_maxrun, k_hist16<2>);
                                  hist16_ms above is then the k_hist16 launch inside it */
    int64_t fallbacks;         /* bit 0: a two-stage attempt was abandoned and the sort-all path ran (deep ties / look-back time-out /
                                  policy decline after the front end); bits 8..: reason code of the abandon */
    int64_t progression_suffixes; /* suffixes finished as arithmetic progressions of positions (tandem repeats, k_chain_resolve) by the
                                  in-place doubling of the last build */
    int64_t bucket_sort_handed_back; /* round 0: segments the fast bucket sort (k_sort_bits / k_sort_fast2) gave to k_sort_mid (length,
                                  key bits or skew outside its shapes); a handful of 65,536 on uniform random bytes up to the class-C limit */
    double hist17_ms;          /* 17-bit histogram of random-like inputs above the class-C limit (part of hist16_ms) */
    int64_t radix_bits;        /* bits of the two scatter levels of the last build: 16, or 17 (level 1 splits 512 ways) */
    int64_t key1_records;      /* small alphabets (<= 84 codes), narrow records: suffixes whose key of the FIRST gather round was read off the
                                  text tile by k_scatter0 and carried through round 0 as a companion word (that round then gathers nothing:
                                  its records are not in gathered_records; round 0 moves 4 bytes more per suffix and pass); 0: every round gathered */
    int64_t doubling_records;  /* single-process sharded / wide builds (logical shards): rows the steps of the distributed prefix doubling scanned, summed
                                  over steps and shards (each: row + group head read, the rank of suffix + h gathered, row + group head written) */
} msufsort_hip_timings;
/* The struct's size is part of the ABI (callers pass timings_out buffers): new fields only ever take reserved slots. */
#ifdef __cplusplus
static_assert(sizeof(msufsort_hip_timings) == 216, "msufsort_hip_timings changed size");
#else
_Static_assert(sizeof(msufsort_hip_timings) == 216, "msufsort_hip_timings changed size");
#endif

int msufsort_hip_device_count(void);
const char* msufsort_hip_strerror(int status);
const char* msufsort_hip_last_error(void);             /* thread-local detail of the last failure */
const char* msufsort_hip_build_id(void);               /* hash of the sources this library was built from (msufsort_amd/csrc/Makefile) */

/* Context: stream + workspace for inputs up to max_n bytes (grown on demand if exceeded). */
int msufsort_hip_ctx_create(msufsort_hip_ctx** out, int32_t device, int64_t max_n);
void msufsort_hip_ctx_destroy(msufsort_hip_ctx* ctx);
void* msufsort_hip_ctx_stream(msufsort_hip_ctx* ctx);  /* hipStream_t, for hipEvent timing */
int msufsort_hip_ctx_sync(msufsort_hip_ctx* ctx);
int msufsort_hip_ctx_trim(msufsort_hip_ctx* ctx);      /* frees the workspace (it is rebuilt on demand) */
/* The host-pointer entry points without a context argument (and msufsort_hip_make_sa_multi) keep their contexts - stream
 * and workspace - for the life of the process, as the reference keeps its worker pool per instance (msufsort.h:311-388);
 * this frees the idle ones. */
void msufsort_hip_release_cached(void);
int msufsort_hip_last_timings(msufsort_hip_ctx* ctx, msufsort_hip_timings* out);

/* ---- suffix array: replaces msufsort::make_suffix_array (reference msufsort.cpp:1730-1767,
 *      first_stage_its cpp:1559-1726 + second_stage_its cpp:1021-1057) ---- */
int msufsort_hip_make_sa_i32(const uint8_t* text, int64_t n, int32_t* sa_out /* n+1 */,
                             const msufsort_hip_opts* opts);
/* Host-pointer variants that reuse a context (stream + workspace) instead of building a temporary one -
 * what a long-lived maniscalco::msufsort instance uses (the reference keeps its worker pool per instance,
 * msufsort.h:311-388). */
int msufsort_hip_make_sa_i32_ctx(msufsort_hip_ctx* ctx, const uint8_t* text, int64_t n, int32_t* sa_out,
                                 const msufsort_hip_opts* opts);
int msufsort_hip_forward_bwt_ctx(msufsort_hip_ctx* ctx, uint8_t* inout, int64_t n, int64_t* sentinel_row,
                                 const msufsort_hip_opts* opts);
int msufsort_hip_inverse_bwt_ctx(msufsort_hip_ctx* ctx, uint8_t* inout, int64_t n, int64_t sentinel_row,
                                 const msufsort_hip_opts* opts);
int msufsort_hip_lcp_i32_ctx(msufsort_hip_ctx* ctx, const uint8_t* text, int64_t n, const int32_t* sa,
                             int32_t* lcp_out);
/* d_text must have at least n + MSUFSORT_HIP_TEXT_PAD readable bytes; the pad is zeroed by the call. */
#define MSUFSORT_HIP_TEXT_PAD 64
int msufsort_hip_make_sa_i32_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                 int32_t* d_sa_out /* n+1 */, const msufsort_hip_opts* opts);
/* Sharded build (multi-GPU): writes only SA[1 + lo, 1 + hi) of the full array into
 * d_slice_out (hi - lo entries) and reports the slice bounds.  Shard 0 also owns SA[0] and the
 * trailing-zero-run rows, which lie inside its slice. */
int msufsort_hip_make_sa_shard_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                   int32_t* d_slice_out, int64_t slice_capacity,
                                   int64_t* slice_lo, int64_t* slice_hi,
                                   const msufsort_hip_opts* opts);
/* Same, for inputs whose ties may run deeper than the key-gather rounds allow (long repeats): also fills
 * d_grp_slice_out[row - lo] = first row of the tie group of that row, RELATIVE to the slice (the row itself when it is
 * final).  Returns 0 when the slice is completely sorted, 1 (MSUFSORT_HIP_UNRESOLVED_GROUPS) when groups remain;
 * *depth_out = number of bytes the members of every remaining group share.  If ANY rank returned 1 the ranks continue
 * with the distributed prefix doubling below.  The _i64 variant runs the wide engine (40-bit indices, int64 rows):
 * any n up to 2^40 - 2 (BASELINE config 5: 8 GiB over 8 GPUs). */
#define MSUFSORT_HIP_UNRESOLVED_GROUPS 1
int msufsort_hip_make_sa_shard_groups_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                          int32_t* d_slice_out, uint32_t* d_grp_slice_out, int64_t slice_capacity,
                                          int64_t* slice_lo, int64_t* slice_hi, int64_t* depth_out,
                                          const msufsort_hip_opts* opts);
int msufsort_hip_make_sa_shard_groups_i64_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                              int64_t* d_slice_out, uint32_t* d_grp_slice_out, int64_t slice_capacity,
                                              int64_t* slice_lo, int64_t* slice_hi, int64_t* depth_out,
                                              const msufsort_hip_opts* opts);
/* Distributed prefix doubling (replaces the reference's tandem-repeat path, msufsort.cpp:316-484; the shards stay the
 * independent sort problems of msufsort.cpp:1652-1683).  State per shard: its slice rows + group heads; the rank array
 * isa[i] = global row of the head of suffix i's group (n + 1 entries of index_bytes = 4 or 8) is replicated and read-only
 * during a step.  One step at offset h (h starts at *depth_out and doubles):
 *   every rank:  msufsort_hip_double_sort_dev      sorts its tied groups by isa[i + h]; rewrites slice rows + group heads
 *                                                  (*emit_items = work items the emit pass has to scan: the context keeps
 *                                                  the list of still-tied rows of its slice between steps, so a step costs
 *                                                  time in proportion to what is tied, not to the slice)
 *                msufsort_hip_emit_updates_dev     work items [i0, i1) of emit_items: rows whose group head changed ->
 *                                                  updates (index_bytes 4: one uint64 = new_row << 32 | suffix; 8: {suffix,
 *                                                  new_row}); windows in ascending order, the one with i1 == items_total
 *                                                  closes the step
 *   exchange:    all-gatherv of the updates (RCCL), then on every rank
 *                msufsort_hip_apply_updates_dev    isa[suffix] = new_row, for the updates of ALL ranks
 * until no rank has tied rows left.  msufsort_hip_isa_from_slice_dev initialises a replica from one (gathered) slice. */
int msufsort_hip_isa_from_slice_dev(msufsort_hip_ctx* ctx, const void* d_sa_slice, const uint32_t* d_grp_slice,
                                    int64_t lo, int64_t hi, void* d_isa, int32_t index_bytes);
int msufsort_hip_double_sort_dev(msufsort_hip_ctx* ctx, int64_t n, void* d_sa_slice, uint32_t* d_grp_slice,
                                 uint32_t* d_grp_prev_slice, int64_t lo, int64_t hi, const void* d_isa, int64_t h,
                                 int32_t index_bytes, const msufsort_hip_opts* opts, int64_t* tied_before, int64_t* emit_items);
int msufsort_hip_emit_updates_dev(msufsort_hip_ctx* ctx, const void* d_sa_slice, const uint32_t* d_grp_slice,
                                  const uint32_t* d_grp_prev_slice, int64_t lo, int64_t hi, int64_t i0, int64_t i1,
                                  int64_t items_total, void* d_updates, int64_t capacity, int32_t index_bytes,
                                  int64_t* count, int64_t* tied_rows);
int msufsort_hip_apply_updates_dev(msufsort_hip_ctx* ctx, const void* d_updates, int64_t count, void* d_isa,
                                   int32_t index_bytes);
/* Two-stage build with a SHARDED first stage (text-like inputs over several GPUs; the reference's own structure, msufsort.cpp:1559-1726
 * + 646-1057, with its first stage - the only part that sorts - split by key range like its partitions, cpp:1652-1683).
 * Every rank calls this with its shard number: the front end (suffix types, histograms) runs on every rank, rank g sorts the B*
 * suffixes of its range of two-byte keys into its slice of d_bstar (ALL sorted B* suffixes, at most n / 2 entries), then
 * `exchange` is called ONCE on every rank - bounds[n_shards + 1] = slice bounds in d_bstar, my_status = 0 (sorted) or 1 (my
 * shard's ties run too deep) - and must (a) agree on one status over all ranks (max), (b) if it is 0, all-gatherv the slices
 * in place (complete on return), (c) return the agreed status (negative: failure).  Then every rank induces the rest from the
 * complete array: d_sa_out holds the WHOLE suffix array on every rank, 4 |B*| = 1.33 n bytes were exchanged instead of 4 n.
 * opts->shard = -1 with exchange = NULL: all shards one after the other on this GPU (logical shards).  opts->two_stage > 0: also
 * below the size / alphabet policy of msufsort_hip_opts.two_stage.
 * Returns 0, MSUFSORT_HIP_TWO_STAGE_DECLINED (every rank alike - the input does not suit the path or some shard's ties ran too
 * deep: take the sharded sort-all path, msufsort_hip_make_sa_shard_groups_dev), MSUFSORT_HIP_TWO_STAGE_FAILED_LOCALLY (this rank
 * only, after the exchange: build the array on this rank alone, msufsort_hip_make_sa_i32_dev with two_stage = -1), or an error. */
#define MSUFSORT_HIP_TWO_STAGE_DECLINED 1
#define MSUFSORT_HIP_TWO_STAGE_FAILED_LOCALLY 2
typedef int (*msufsort_hip_exchange_fn)(void* user, const int64_t* bounds, int32_t n_shards, int32_t my_status);
int msufsort_hip_make_sa_two_stage_sharded_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n, int32_t* d_sa_out /* n+1 */,
                                               uint32_t* d_bstar, int64_t bstar_capacity, msufsort_hip_exchange_fn exchange, void* user,
                                               const msufsort_hip_opts* opts);
/* Slice bounds only (all shards), without sorting: bounds[n_shards + 1], in SA rows. */
int msufsort_hip_shard_bounds_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                  int32_t n_shards, int64_t* bounds);

/* The 16-bit histogram computed SHARDED (SURVEY.md section 8(e) "Partitioning": "if computed sharded: one all-reduce" of
 * 65,536 counters; the reference counts per thread and sums, count_suffixes msufsort.cpp:1496-1521, :1603-1630).  A shard
 * build (msufsort_hip_make_sa_shard_*_dev) otherwise starts with a pass over the WHOLE text on every rank.  Three calls per
 * build and rank, with two small collectives of the caller's in between (msufsort_amd/dist.py: plan_sharded):
 *   1. hist_part:  counts the two-byte keys of the suffixes that start in this part's scatter stripes of the text (stripes
 *      [first, end) of `total`: stripes_out[0..2] = total, first, end; part p of P owns stripes [total p / P, total (p + 1) / P))
 *      and writes the part's totals to d_hist_out (65,536 x uint64, key T[i] << 8 | T[i+1]).    -> all-reduce (SUM) of d_hist_out
 *   2. hist_plan:  plans the n_shards key ranges from the summed histogram exactly as msufsort_hip_shard_bounds_dev would
 *      (bounds_out[n_shards + 1], suffix-array rows) and writes, for EVERY shard g, the first-byte counts of my stripes over g's key
 *      range - what g's scatter needs to place its records: d_sums_out[g][stripes_per_part][256] (uint32; rows beyond my stripe
 *      count stay zero).  Returns MSUFSORT_HIP_HIST_NEEDS_REPLICA when a shard boundary has to fall INSIDE a heavy two-byte key
 *      (DNA, text: the plan then needs a deeper histogram of that key): nothing is kept, the shard build computes its own
 *      histogram as before - on every rank alike, since all ranks hold the same sum.      -> all-gather of d_sums_out
 *   3. hist_install: the counts of ALL stripes over MY shard's key range, d_stripe_sums[total][256], assembled from the gathered
 *      blocks (stripes in text order).  The next msufsort_hip_make_sa_shard_*_dev call for this text, n_shards and shard runs
 *      without a histogram pass; any other call drops the state. */
#define MSUFSORT_HIP_HIST_NEEDS_REPLICA 3
int msufsort_hip_hist_part_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n, int32_t part, int32_t parts,
                               uint64_t* d_hist_out /* 65536 */, int32_t* stripes_out /* 3, may be NULL */);
int msufsort_hip_hist_plan_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n, int32_t n_shards, const uint64_t* d_hist_sum,
                               uint32_t* d_sums_out, int32_t stripes_per_part, int64_t* bounds_out);
int msufsort_hip_hist_install_dev(msufsort_hip_ctx* ctx, int32_t shard, const uint32_t* d_stripe_sums, int32_t stripes);

/* 64-bit output (callers with 64-bit index types; the reference's suffix_index is int32 with two
 * flag bits, msufsort.h:47,84-93, i.e. n < 2^30).  n <= 2^31 - 2: the int32 rows, widened on the device.  Larger inputs, up to
 * 2^40 - 2 bytes: the wide engine - 8-byte records (24-bit key, 40-bit index), as many logical shards as the workspace
 * needs taking turns on the context's GPU, int64 rows written in place, distributed prefix doubling for deep ties. */
int msufsort_hip_make_sa_i64(const uint8_t* text, int64_t n, int64_t* sa_out /* n+1 */,
                             const msufsort_hip_opts* opts);
int msufsort_hip_make_sa_i64_ctx(msufsort_hip_ctx* ctx, const uint8_t* text, int64_t n, int64_t* sa_out,
                                 const msufsort_hip_opts* opts);
int msufsort_hip_make_sa_i64_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                 int64_t* d_sa_out /* n+1 */, const msufsort_hip_opts* opts);

/* ---- one process, several GPUs (SURVEY 8(b) "device list / count ... defaulting to all visible GPUs", 8(e)) ----
 * Host text in, host suffix array out (n + 1 entries of index_bytes = 4, or 8 for inputs beyond 2^31 - 2 bytes / force_wide).
 * devices == NULL or n_dev == 0: the list in the environment variable MSUFSORT_DEVICES ("0,1,..."), else all visible GPUs.
 * One host thread per device; every device sorts its key ranges and each
 * finished slice leaves for the host while the next one is sorted (with ONE device this is the streaming host path: the
 * D2H of 4(n+1) bytes overlaps the remaining sorts).  Deep ties: distributed prefix doubling, rank updates exchanged with
 * peer-to-peer copies over xGMI.  opts->n_shards: total number of key-range shards (default: 8 per device from 64 MiB on, 16 from 512 MiB).
 * Replaces the cost the reference pays at msufsort.cpp:1754-1758 (allocation + first touch of the result). */
int msufsort_hip_make_sa_multi(const int32_t* devices, int32_t n_dev, const uint8_t* text, int64_t n, void* sa_out,
                               int32_t index_bytes, const msufsort_hip_opts* opts, msufsort_hip_timings* timings_out);

/* Host-only helper used by the two calls above (no device work): balanced key-range cuts from the
 * exclusive prefix bstart[65537] of the 16-bit histogram; cuts/rows have n_shards+1 entries. */
int msufsort_hip_plan_cuts(const uint64_t* bstart, int64_t n, int64_t z, int32_t n_shards,
                           uint32_t* cuts, int64_t* rows);

/* ---- forward BWT: replaces msufsort::forward_burrows_wheeler_transform (cpp:1771-1817) ----
 *      (n > 2^31 - 2, or opts->force_wide: through the wide engine and int64 rows) */
int msufsort_hip_forward_bwt(uint8_t* inout, int64_t n, int64_t* sentinel_row,
                             const msufsort_hip_opts* opts);
/* d_bwt_out must not overlap d_text (BAD_ARG): the bytes are written while the text is still being read - after a two-stage build
 * (text-like inputs) region by region beside the remaining induction levels, on the context's second stream; the call returns when
 * both streams have finished. */
int msufsort_hip_forward_bwt_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                 uint8_t* d_bwt_out /* n */, int64_t* sentinel_row,
                                 const msufsort_hip_opts* opts);
/* BWT from an SA that is already in HBM (gather T[SA[r]-1]). */
int msufsort_hip_bwt_from_sa_dev(msufsort_hip_ctx* ctx, const uint8_t* d_text, int64_t n,
                                 const int32_t* d_sa, uint8_t* d_bwt_out, int64_t* sentinel_row);
int msufsort_hip_bwt_from_sa_i64_dev(msufsort_hip_ctx* ctx, const uint8_t* d_text, int64_t n,
                                     const int64_t* d_sa, uint8_t* d_bwt_out, int64_t* sentinel_row);

/* Sharded forward transform (SURVEY 8(e) "Collective: ... for BWT gather n/G-byte slices instead"; the reference writes its n
 * output bytes in place and returns the sentinel row, cpp:1771-1817).  For the FINISHED slice rows [lo, hi) of a sharded build:
 * d_row_bytes_out[r - lo] = T[SA[r] - 1] (one byte per row; the row of suffix 0 gets a placeholder), *sentinel_row = that row
 * if it lies in the slice, else -1.  The ranks agree on the sentinel row (max), every rank moves its bytes to
 * out[r - (r > sentinel)] and ONE all-gatherv of the byte slices completes the transform: n bytes travel instead of the
 * 4(n+1) or 8(n+1) of the rows (msufsort_amd/dist.py::forward_bwt_sharded). */
int msufsort_hip_bwt_slice_dev(msufsort_hip_ctx* ctx, const uint8_t* d_text, int64_t n, const void* d_sa_slice, int64_t lo, int64_t hi,
                               int32_t index_bytes, uint8_t* d_row_bytes_out, int64_t* sentinel_row);

/* One process, the listed devices (as msufsort_hip_make_sa_multi: NULL / 0 = MSUFSORT_DEVICES, else all visible GPUs), host bytes
 * in place: key-range shards, and what leaves a device is the BYTE in front of every suffix of a finished slice - n bytes over PCIe
 * instead of 4 (n + 1), streamed while the remaining shards are sorted (nothing lands in the caller's buffer before every device has
 * read the text from it).  Text-like and small inputs take msufsort_hip_forward_bwt on the first device.
 * On error the contents of `inout` are UNDEFINED: byte slices of shards that finished before a device failed have already replaced
 * the text there (workers stop queuing copies as soon as any device has failed; the single-device msufsort_hip_forward_bwt and the
 * reference write nothing before the whole transform exists). */
int msufsort_hip_forward_bwt_multi(const int32_t* devices, int32_t n_dev, uint8_t* inout, int64_t n, int64_t* sentinel_row,
                                   const msufsort_hip_opts* opts, msufsort_hip_timings* timings_out);

/* ---- inverse BWT: replaces msufsort::reverse_burrows_wheeler_transform (cpp:1821-2096) ---- */
int msufsort_hip_inverse_bwt(uint8_t* inout, int64_t n, int64_t sentinel_row,
                             const msufsort_hip_opts* opts);
int msufsort_hip_inverse_bwt_dev(msufsort_hip_ctx* ctx, const uint8_t* d_bwt, int64_t n,
                                 int64_t sentinel_row, uint8_t* d_text_out,
                                 const msufsort_hip_opts* opts);

/* ---- LCP in the demo's convention (reference src/executable/msufsort/main.cpp:16-159) ---- */
int msufsort_hip_lcp_i32(const uint8_t* text, int64_t n, const int32_t* sa /* n+1 */,
                         int32_t* lcp_out /* n */, const msufsort_hip_opts* opts);
int msufsort_hip_lcp_i32_dev(msufsort_hip_ctx* ctx, const uint8_t* d_text, int64_t n,
                             const int32_t* d_sa, int32_t* d_lcp_out);

/* ---- validation on device: checker of reference main.cpp:236-270 (+ permutation check) ---- */
int msufsort_hip_validate_sa_dev(msufsort_hip_ctx* ctx, const uint8_t* d_text, int64_t n,
                                 const int32_t* d_sa, int64_t* error_count);
int msufsort_hip_validate_sa_i64_dev(msufsort_hip_ctx* ctx, const uint8_t* d_text, int64_t n,
                                     const int64_t* d_sa, int64_t* error_count);

/* ---- stage probes used by the parity tests (not part of the reference surface) ---- */
int msufsort_hip_debug_hist16_dev(msufsort_hip_ctx* ctx, uint8_t* d_text, int64_t n,
                                  uint32_t* d_hist /* 65536 */);

#ifdef __cplusplus
}
#endif
#endif /* MSUFSORT_HIP_H */
