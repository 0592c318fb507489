"""Multi-GPU path: one process per GPU, the 16-bit key space is split into
`world` count-balanced contiguous ranges, every rank radix-sorts its range into its own slice of the
full suffix array and the slices are exchanged with ONE all-gatherv.  RCCL has no v-variant, so the
gather is ONE group of point-to-point sends/receives (each peer sends its slice directly to every other peer,
using all xGMI links at once) - torch.distributed is plumbing here, the sort is the HIP path.

The reference has no distributed code at all; this is the MI355X-native extension
of its bucket-parallel first stage (reference msufsort.cpp:1652-1683 hands partitions to threads).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _many(world: int) -> bool:
    """Are the collectives of a step issued?  Yes with more than one rank - and, under MSUFSORT_DIST_ALWAYS_COLLECTIVE=1, with one
    rank too: the hook that lets a one-GPU box put every all-reduce / all-gatherv call of this module through RCCL itself
    (tests/test_gpu_dist.py::test_bench_dist_path_on_rccl_with_one_rank; a one-rank all-gatherv posts no sends)."""
    import os
    return world > 1 or bool(os.environ.get("MSUFSORT_DIST_ALWAYS_COLLECTIVE"))


def plan_cuts(bstart, n: int, z: int, n_shards: int):
    """Host-only: (cuts, rows) for `n_shards` shards from the exclusive 16-bit-key prefix bstart[65537] (uint64)."""
    b = np.ascontiguousarray(bstart, dtype=np.uint64)
    assert b.size == 65537
    cuts = np.zeros(n_shards + 1, dtype=np.uint32)
    rows = np.zeros(n_shards + 1, dtype=np.int64)
    _lib.check(_lib.lib().msufsort_hip_plan_cuts(b.ctypes.data, n, z, n_shards, cuts.ctypes.data, rows.ctypes.data), "plan_cuts")
    return cuts.tolist(), rows.tolist()


_MODE = {"mode": "p2p"}


def select_exchange(dist, device):
    """Picks the all-gatherv flavour once per process group: grouped point-to-point (default) unless
    MSUFSORT_ALLGATHERV=bcast is set or a tiny trial exchange fails on ANY rank (all ranks then agree on bcast)."""
    import os
    import torch
    if os.environ.get("MSUFSORT_ALLGATHERV", "p2p") == "bcast":
        _MODE["mode"] = "bcast"
        return "bcast"
    world, rank = dist.get_world_size(), dist.get_rank()
    # (a probe that RAISES aborts the job: swallowing it here would leave the peers blocked in the grouped exchange)
    probe = torch.zeros(world * 4, dtype=torch.int32, device=device)
    probe[rank * 4:(rank + 1) * 4] = rank + 1
    _MODE["mode"] = "p2p"
    allgatherv_slices(probe, [4 * g for g in range(world + 1)], dist)
    ok = int(bool((probe.view(world, 4) == torch.arange(1, world + 1, device=device, dtype=torch.int32)[:, None]).all()))
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    _MODE["mode"] = "p2p" if int(flag.item()) == 1 else "bcast"
    return _MODE["mode"]


def allgatherv_ranges(full, ranges, dist, group=None, wait=True):
    """The grouped exchange behind allgatherv_slices for arbitrary row ranges: rank g has filled full[ranges[g][0]:ranges[g][1]]
    (ranges need not be adjacent: the j-th sub-slices of all ranks, posted while the next sub-shards are still being sorted)."""
    world = len(ranges)
    rank = dist.get_rank(group)
    if _MODE["mode"] == "bcast":
        works = [dist.broadcast(full[ranges[g][0]:ranges[g][1]], src=g, group=group, async_op=True) for g in range(world) if ranges[g][1] > ranges[g][0]]
    else:
        ops = []
        lo, hi = ranges[rank]
        for step in range(1, world):
            dst = (rank + step) % world
            src = (rank - step) % world
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, full[lo:hi], dst, group))
            if ranges[src][1] > ranges[src][0]:
                ops.append(dist.P2POp(dist.irecv, full[ranges[src][0]:ranges[src][1]], src, group))
        works = dist.batch_isend_irecv(ops) if ops else []
    if not wait:
        return works
    wait_all(works, full)
    return full


def sub_shards_for(world: int, n: int, gather_rows: bool = True) -> int:
    """Sub-shards a rank cuts its key range into so that finished sub-slices travel while the next sub-shard is sorted (round 6; the
    reference's threads pop partitions and leave their rows while others still sort, msufsort.cpp:1652-1683).  Every sub-shard reads
    the whole text once more in its level-0 scatter (~0.3 ms per GiB) and costs ~0.1 ms of host round trips, the histogram and the
    plan are shared (msufsort_hip_opts.reuse_plan): four where rows are exchanged and the input is at least 64 MiB, else one.
    MSUFSORT_DIST_SUBSHARDS=k forces k (tests: small inputs)."""
    import os
    e = int(os.environ.get("MSUFSORT_DIST_SUBSHARDS", "0") or 0)
    if e > 0:
        return e
    return 4 if (world > 1 and gather_rows and n >= (64 << 20)) else 1


def allgatherv_slices(full, bounds, dist, group=None, wait=True):
    """All-gatherv of SA slices IN PLACE: rank g has filled full[bounds[g]:bounds[g+1]]; afterwards every rank
    holds the whole array.  RCCL has no v-variant: every rank posts, as ONE group (ncclGroupStart/End through
    batch_isend_irecv), a send of its slice to each peer and a receive of each peer's slice straight into its
    place - all xGMI links of the fully connected node carry one slice each way at the same time (per-root
    broadcasts issued one after another would serialise on the communicator's stream).
    wait=False returns the pending work handles (finish them with `wait_all`) so that the next build can run
    while the links are busy."""
    world = len(bounds) - 1
    rank = dist.get_rank(group)
    if _MODE["mode"] == "bcast":                     # fallback: one broadcast per root (serialised on the communicator)
        works = [dist.broadcast(full[bounds[g]:bounds[g + 1]], src=g, group=group, async_op=True)
                 for g in range(world) if bounds[g + 1] > bounds[g]]
        if not wait:
            return works
        wait_all(works, full)
        return full
    ops = []
    lo, hi = bounds[rank], bounds[rank + 1]
    for step in range(1, world):
        dst = (rank + step) % world
        src = (rank - step) % world
        if hi > lo:
            ops.append(dist.P2POp(dist.isend, full[lo:hi], dst, group))
        if bounds[src + 1] > bounds[src]:
            ops.append(dist.P2POp(dist.irecv, full[bounds[src]:bounds[src + 1]], src, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if not wait:
        return works
    wait_all(works, full)
    return full


def wait_all(works, full=None):
    """Completes pending exchange work.  work.wait() only orders torch's current stream behind the communicator's;
    the engine runs on its own HIP stream, so the device is synchronised before anything else touches the rows."""
    for w in works:
        w.wait()
    if full is not None and full.is_cuda:
        import torch
        torch.cuda.synchronize(full.device)


def _window_rows(default: int) -> int:
    """Rows per exchange window (rank updates, group heads).  MSUFSORT_DIST_WINDOW shrinks it: a test hook that makes small
    inputs walk several windows."""
    import os
    v = int(os.environ.get("MSUFSORT_DIST_WINDOW", "0") or 0)
    return max(1, v) if v > 0 else default


class ShardState:
    """Per-rank buffers of the distributed prefix doubling (allocated on first use, reused by later builds).
    Sizes for n = 2^33 over 8 ranks (BASELINE config 5; rows_max = 2^30): rank replica 8(n+2) = 64 GiB, grp_prev 4 GiB,
    update windows 0.5 + 4 GiB, group-head windows 0.25 + 2 GiB (DESIGN.md section 3.7 has the whole budget).
    MSUFSORT_DIST_WINDOW (test hook) must be set alike on every rank: the windows size the receive buffers, and
    _distributed_doubling checks that the ranks agree before anything travels."""

    def __init__(self):
        self.isa = None
        self.grp_prev = None
        self.upd_local = None
        self.upd_all = None
        self.grp_all = None
        self.stats = {}

    @staticmethod
    def windows(rows_max, index_bytes):
        """(rows per update window, rows per group-head window, 64-bit words per update) - the one place both ensure() and
        bytes_needed() take them from."""
        e = 2 if index_bytes == 8 else 1          # one update = ONE 64-bit word for int32 rows (new_row << 32 | suffix), TWO for int64 rows
        return max(1, min(rows_max, _window_rows(1 << 25))), max(1, min(rows_max, _window_rows(1 << 26))), e

    @staticmethod
    def bytes_needed(n, rows_max, world, index_bytes):
        """What ensure() allocates (bench.py's memory check)."""
        win, gwin, e = ShardState.windows(rows_max, index_bytes)
        return (n + 2) * index_bytes + max(rows_max, 1) * 4 + win * e * 8 * (world + 1) + gwin * 4 * world

    def ensure(self, n, rows_max, world, index_bytes, device):
        import torch
        dt = torch.int64 if index_bytes == 8 else torch.int32
        if self.isa is None or self.isa.numel() < n + 2 or self.isa.dtype != dt:
            self.isa = None                      # (drop the old replica before the new one is allocated)
            self.isa = torch.empty(n + 2, dtype=dt, device=device)
        if self.grp_prev is None or self.grp_prev.numel() < rows_max:
            self.grp_prev = torch.empty(max(rows_max, 1), dtype=torch.int32, device=device)
        self.win, self.gwin, e = ShardState.windows(rows_max, index_bytes)
        if self.upd_local is None or self.upd_local.numel() < self.win * e:
            self.upd_local = torch.empty(self.win * e, dtype=torch.int64, device=device)
        if self.upd_all is None or self.upd_all.numel() < self.win * e * world:
            self.upd_all = torch.empty(self.win * e * world, dtype=torch.int64, device=device)
        if self.grp_all is None or self.grp_all.numel() < self.gwin * world:
            self.grp_all = torch.empty(self.gwin * world, dtype=torch.int32, device=device)


def update_offsets(counts, index_bytes):
    """Element offsets (int64 words) of every rank's updates inside one exchange window: counts[g] updates of rank g, one word
    each for int32 rows, two for int64 rows.  Returns (pre, e): rank g's updates occupy words pre[g] .. pre[g + 1]."""
    e = 2 if index_bytes == 8 else 1
    pre = [0]
    for x in counts:
        pre.append(pre[-1] + int(x) * e)
    return pre, e


def _replicate_ranks(ctx, d_sa_full, d_grp, bounds, rank, world, dist, state, index_bytes):
    """Builds this rank's replica of the rank array from the gathered provisional rows and the group heads of EVERY slice.
    The group heads (uint32, relative to their slice) stay distributed: they travel in windows of `gwin` rows per rank - one
    all-gatherv per window - instead of as one 4(n+1)-byte array per GPU (32 GiB at n = 2^33)."""
    lo, hi = bounds[rank], bounds[rank + 1]
    rows = [bounds[g + 1] - bounds[g] for g in range(world)]
    W = state.gwin
    stage = state.grp_all
    for w0 in range(0, max(rows), W):
        cnt = [max(0, min(W, r - w0)) for r in rows]
        pre = [0]
        for x in cnt:
            pre.append(pre[-1] + x)
        if cnt[rank]:
            stage[pre[rank]:pre[rank + 1]] = d_grp[w0:w0 + cnt[rank]]
        if stage.is_cuda:
            # the copy above ran on torch's stream; the exchange runs on the communicator's, isa_from_slice on the engine's
            import torch
            torch.cuda.current_stream(stage.device).synchronize()
        if _many(world):
            allgatherv_slices(stage, pre, dist)
        for g in range(world):
            if cnt[g]:
                r0 = bounds[g] + w0
                # (group heads are relative to their SLICE: the base row is the slice's first row, whatever the window)
                ctx.isa_from_slice(d_sa_full[r0:r0 + cnt[g]], stage[pre[g]:pre[g + 1]], bounds[g], bounds[g] + cnt[g], state.isa, index_bytes)


def _distributed_doubling(ctx, n, d_sa_full, d_grp, bounds, rank, world, dist, depth, index_bytes, state, verbose=0):
    """Prefix doubling over the shards (include/msufsort_hip.h, 'Distributed prefix doubling'): every rank sorts only the
    tie groups of its own slice; the rank array is replicated and refreshed once per step with ONE all-gatherv of the
    (suffix, new head row) updates of all ranks.  Starts from gathered provisional rows (d_sa_full, all slices) and THIS rank's
    group heads (d_grp: one uint32 per row of my slice, relative to the slice).  index_bytes = 8: int64 rows and ranks, 16-byte
    updates (BASELINE config 5; the reference's int32 index with two flag bits stops at 2^30, msufsort.h:47, 84-93)."""
    import time

    import torch
    dev = d_sa_full.device
    lo, hi = bounds[rank], bounds[rank + 1]
    rows_max = max(bounds[g + 1] - bounds[g] for g in range(world))
    state.ensure(n, rows_max, world, index_bytes, dev)
    isa = state.isa
    one = torch.empty(1, dtype=d_sa_full.dtype, device=dev)
    sl = d_sa_full[lo:hi] if hi > lo else one
    gl = d_grp[:hi - lo] if hi > lo else torch.empty(1, dtype=torch.int32, device=dev)
    gp = state.grp_prev
    if _many(world):
        # the exchange windows size every rank's receive buffers: the ranks must have derived the same ones (MSUFSORT_DIST_WINDOW is
        # read per process)
        w = torch.tensor([state.win, -state.win, state.gwin, -state.gwin], dtype=torch.int64, device=dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        w = w.tolist()
        if w[0] != -w[1] or w[2] != -w[3]:
            raise _lib.MsufsortHipError(f"exchange windows differ between ranks (updates {-w[1]} .. {w[0]}, group heads {-w[3]} .. {w[2]}): set MSUFSORT_DIST_WINDOW alike everywhere")
    _replicate_ranks(ctx, d_sa_full, d_grp, bounds, rank, world, dist, state, index_bytes)
    st = {"doubling_steps": 0, "sort_ms": 0.0, "exchange_ms": 0.0, "updates": 0, "depth": depth, "index_bytes": index_bytes, "windows": 0}
    win = state.win
    h = depth
    live = hi > lo
    while True:
        items = 0
        if live:
            tied_before, items = ctx.double_sort(n, sl, gl, gp, lo, hi, isa, h, index_bytes, verbose)
            live = tied_before > 0
            st["sort_ms"] += ctx.timings().refine_ms
        nw = torch.tensor([(items + win - 1) // win if live else 0], dtype=torch.int64, device=dev)
        if _many(world):
            dist.all_reduce(nw, op=dist.ReduceOp.MAX)        # every rank walks the same number of exchange windows
        nwin = int(nw.item())
        tied_local = 0
        for w in range(nwin):
            cnt = tied = 0
            if live and w * win < items:
                i0 = w * win
                i1 = min(i0 + win, items)
                cnt, tied = ctx.emit_updates(sl, gl, gp, lo, hi, i0, i1, items, state.upd_local, win, index_bytes)
            tied_local += tied
            t0 = time.perf_counter()
            counts = torch.zeros(world, dtype=torch.int64, device=dev)
            counts[rank] = cnt
            if _many(world):
                dist.all_reduce(counts)
            pre, e = update_offsets(counts.tolist(), index_bytes)
            if pre[-1]:
                if cnt:
                    state.upd_all[pre[rank]:pre[rank + 1]] = state.upd_local[:cnt * e]
                if dev.type == "cuda":
                    torch.cuda.current_stream(dev).synchronize()
                if _many(world):
                    allgatherv_slices(state.upd_all, pre, dist)
                st["exchange_ms"] += (time.perf_counter() - t0) * 1e3
                ctx.apply_updates(state.upd_all, pre[-1] // e, isa, index_bytes)
                st["updates"] += pre[-1] // e
            st["windows"] += 1
        tt = torch.tensor([tied_local], dtype=torch.int64, device=dev)
        if _many(world):
            dist.all_reduce(tt)
        st["doubling_steps"] += 1
        if int(tt.item()) == 0:
            break
        h *= 2
        if h > 2 * n + 2:
            raise _lib.MsufsortHipError("distributed prefix doubling did not converge")
    state.stats = st
    return st


def build_sa_two_stage_sharded(ctx, d_text, n: int, d_sa_full, d_bstar, rank: int, world: int, dist, two_stage: int = 0, verbose: int = 0, stats=None):
    """Text-like inputs over several GPUs the way the reference structures its build (msufsort.cpp:1559-1726 + 646-1057): only the
    B* suffixes are sorted - sharded by key range, rank g its own shard - their slices are all-gathered (4 |B*| = 1.33 n bytes, a
    third of the suffix array), and every rank induces all other suffixes from the complete sorted-B* array: the WHOLE array
    ends up on every rank without any further exchange.  d_bstar: int32 scratch of at least n // 2 + 1 entries.
    Returns True when d_sa_full is complete; False when the path declined on EVERY rank (not text-like / too small / ties too
    deep): the caller continues with build_sa_sharded.  A rank whose induction fails after the exchange rebuilds locally."""
    import time

    import torch
    dev = d_sa_full.device
    t_ex = [0.0]

    def exchange(bounds, my_status):
        t0 = time.perf_counter()
        flag = torch.tensor([my_status], dtype=torch.int32, device=dev)
        if _many(world):
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        agreed = int(flag.item())
        if agreed == 0 and _many(world):
            allgatherv_slices(d_bstar, bounds, dist)          # waits, and synchronises the device (wait_all)
        t_ex[0] = (time.perf_counter() - t0) * 1e3
        return agreed

    r = ctx.make_sa_two_stage_sharded(d_text, n, d_sa_full, d_bstar, rank, world, exchange, two_stage=two_stage, verbose=verbose)
    if stats is not None:
        stats["two_stage_status"] = r
        stats["bstar_exchange_ms"] = round(t_ex[0], 3)
    if r == 2:          # look-back time-out on this rank only: the others are done, nothing collective is left
        ctx.make_sa(d_text, n, d_sa_full, two_stage=-1)
        return True
    return r == 0


def sharded_hist_enabled(world: int, n: int = None) -> bool:
    """Does a sharded build start with plan_sharded?  MSUFSORT_DIST_SHARDED_HIST=1 / 0: always / never.  Default: where it pays -
    the replicated histogram costs 0.30 ms per GiB of text on every rank, the sharded one 1/world of that plus ~0.17 ms of fixed
    cost more than the replicated plan's (three calls with a stream synchronisation each; measured per rank, collectives not
    included, tools/gpu_sharded_hist.py: 256 MiB / 8 ranks +0.03 ms, 1 GiB / 8 ranks -0.13 ms, 2 GiB / 8 ranks -0.40 ms) and two
    small collectives (~0.1 ms): from 0.75 GiB of text counted by OTHER ranks (8 ranks: n >= 0.86 GiB; 4: 1 GiB; 2: 1.5 GiB).
    n = None: could any size take it?"""
    import os
    e = os.environ.get("MSUFSORT_DIST_SHARDED_HIST", "")
    if not _many(world) or e == "0":
        return False
    return e == "1" or n is None or n * (world - 1) >= world * (3 << 28)


def plan_sharded(ctx, d_text, n: int, rank: int, world: int, dist, stats=None, group=None, device=None, sub: int = 1):
    """The 16-bit histogram of a sharded build computed SHARDED (SURVEY.md section 8(e) "Partitioning"): every rank counts 1/world
    of the text's scatter stripes, ONE all-reduce (512 KiB) gives everybody the totals, every rank plans the same key ranges
    from them, and ONE all-gather (128 KiB per rank) hands every shard the per-stripe first-byte counts of its key range that
    its scatter needs from the stripes the others counted.  The shard build that follows on this context then starts without
    a pass over the whole text (reference: per-thread counts summed, msufsort.cpp:1496-1521, :1603-1630).
    Returns the slice bounds, or None when the plan needs a boundary inside a heavy two-byte key (DNA, text): every rank gets the
    same answer (same totals), nothing is kept and the shard builds compute their own histogram as before.
    sub > 1: the plan is made for world * sub shards (rank g sorts the shards g * sub .. g * sub + sub - 1 one after the other, its
    finished sub-slices travelling meanwhile); returns (bounds of the world * sub shards, the stripe sums of my sub-shards - the first
    is installed, the caller installs the next one before it builds that sub-shard)."""
    import torch
    if n < 1:
        return None
    dev = device if device is not None else d_text.device

    def sync():          # the collectives run on the communicator's stream, torch's copies on torch's, the C calls on the engine's
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()

    # A rank-local failure between the collectives (an allocation, a C call) must not leave the peers blocked in the next one:
    # every rank's status travels WITH the data (a spare word of the totals; a 4-byte all-reduce before the all-gather) and all
    # ranks leave together - the same relay the two-stage exchange has (build_sa_two_stage_sharded).
    h = torch.zeros(65537, dtype=torch.int64, device=dev)          # [65536]: ranks that failed counting their part
    total = 0
    err = None
    try:
        total, s0, s1 = ctx.hist_part(d_text, n, rank, world, h[:65536])
    except Exception as e:  # noqa: BLE001
        err = e
        h.zero_()
        h[65536] = 1
    if total == 0 and err is None:          # (nothing but zero bytes: every rank sees that alike)
        return None
    if _many(world):
        sync()
        dist.all_reduce(h, group=group)
    sync()                                   # hist_plan reads the reduced totals on the engine's stream
    if err is not None or int(h[65536].item()):
        raise err if err is not None else _lib.MsufsortHipError("sharded histogram: a peer failed counting its stripes")
    per = max(1, -(-total // world))
    nsh = world * max(1, sub)
    bounds, sums = None, None
    status = 0                               # 0 planned, 1 the plan needs a replicated histogram (every rank alike), 2 failed here
    try:
        sums = torch.empty((nsh, per, 256), dtype=torch.int32, device=dev)
        bounds = ctx.hist_plan(d_text, n, nsh, h[:65536], sums)
        if bounds is None:
            status = 1
    except Exception as e:  # noqa: BLE001
        err, status = e, 2
    if _many(world):
        flag = torch.tensor([status], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        agreed = int(flag.item())
    else:
        agreed = status
    if agreed == 2:
        raise err if err is not None else _lib.MsufsortHipError("sharded histogram: a peer failed planning the shards")
    if agreed == 1:
        return None
    if world > 1:
        got = [torch.empty_like(sums) for _ in range(world)]
        sync()
        dist.all_gather(got, sums, group=group)
    else:
        if _many(world):
            sync()
            dist.all_gather([torch.empty_like(sums)], sums, group=group)        # (the one-rank RCCL hook: the collective is issued all the same)
        got = [sums]
    # part p counted the stripes [total p / world, total (p + 1) / world): its block for each of MY shards, in text order
    mine = [torch.cat([got[p][rank * max(1, sub) + j, :(total * (p + 1) // world - total * p // world)] for p in range(world)]).contiguous()
            for j in range(max(1, sub))]
    sync()                                   # (the gathered blocks assembled on torch's stream before the engine's stream reads them)
    try:
        ctx.hist_install(rank * max(1, sub), mine[0])
    except _lib.MsufsortHipError:
        # local and recoverable: nothing collective depends on it - this rank's shard builds count for themselves (same plan: same totals)
        return bounds if sub <= 1 else (bounds, None)
    if stats is not None:
        stats["sharded_hist"] = stats.get("sharded_hist", 0) + 1
    return bounds if sub <= 1 else (bounds, mine)


def build_sa_sharded(ctx, d_text, n: int, d_sa_full, rank: int, world: int, dist, bounds=None, text_rounds: int = 0,
                     d_grp=None, overlap=False, index_bytes: int = 4, state=None, verbose: int = 0, gather_rows: bool = True, stats=None,
                     hist_group=None, sub_bounds=None):
    """One step of the sharded build on this rank: sort my key range into my slice, all-gatherv the slices.

    Deep ties (long repeats) cannot be finished by key gathers.  With `d_grp` (int32 view of uint32, at least as many entries
    as my slice has rows - NOT the whole array) every rank also keeps the tie groups of its slice; if any rank stopped with
    unresolved groups the provisional rows are gathered once, the group heads travel in windows, every rank builds its replica
    of the rank array, and the ranks run the DISTRIBUTED prefix doubling: each sorts only its own groups, one all-gatherv of
    rank updates per step.  The final slices are gathered at the end.  Without `d_grp` such inputs raise
    (MSUFSORT_HIP_ERR_UNSUPPORTED).
    index_bytes = 8: wide engine (int64 rows; any n up to 2^40 - 2; needs d_grp) - BASELINE config 5.
    text_rounds: key-gather rounds before the shards hand over to the doubling (0: 8, wide engine 3 - what the single-process
    driver msufsort_hip_make_sa_multi uses).
    gather_rows=False: the rows stay distributed (every rank's slice d_sa_full[bounds[rank]:bounds[rank+1]] is final on return;
    the other slices hold provisional rows or nothing) - for consumers that exchange something smaller, like forward_bwt_sharded.

    sub_bounds (round 6; plan_sub_bounds): the row bounds of world * k shards - rank g sorts the k SUB-SHARDS g k .. g k + k - 1 of its
    key range one after the other and posts sub-slice j (one group of direct sends / receives, like the whole slice before) as soon
    as it is sorted: the links carry it while sub-shard j + 1 is sorted on the engine's stream (the reference's workers leave
    their partitions while others still sort, msufsort.cpp:1652-1683).  `bounds` must then be sub_bounds[::k].  The return value and
    the latency semantics are unchanged: without overlap=True the call returns when every rank holds every row.

    overlap=True: returns the pending exchange handles instead of waiting, so the caller can start the next build
    (into ANOTHER output buffer) while the slices travel; finish with `wait_all(works, d_sa_full)`.  If the build
    turns out to need the doubling phase everything is completed here and [] is returned.

    The build starts with plan_sharded where that pays (sharded_hist_enabled): the histogram counted 1/world per rank.  hist_group: a
    process group of its own for those two small collectives (and for the 16-byte agreement on the shards' status) - they would
    otherwise queue behind slices on the communicator's stream; stats["sharded_hist"] counts the builds that started this way."""
    import torch
    k = 1
    if sub_bounds is not None:
        if (len(sub_bounds) - 1) % world:
            raise ValueError("sub_bounds must hold the bounds of world * k shards")
        k = (len(sub_bounds) - 1) // world
        if bounds is not None and list(bounds) != list(sub_bounds[::k]):
            raise ValueError("bounds must be sub_bounds[::k]")
        bounds = list(sub_bounds[::k])
    installs = None
    if sharded_hist_enabled(world, n) and hasattr(ctx, "hist_part"):
        # the histogram this build starts with, counted 1/world per rank (two small collectives; the shard builds below consume it)
        planned = plan_sharded(ctx, d_text, n, rank, world, dist, stats=stats, group=hist_group, device=d_sa_full.device, sub=k)
        if planned is not None:
            if k > 1:
                planned, installs = planned
                if list(sub_bounds) != list(planned):
                    raise _lib.MsufsortHipError("the sub-shard bounds passed in are not the plan of this text")
            else:
                if bounds is not None and list(bounds) != planned:
                    raise _lib.MsufsortHipError("the bounds passed in are not the plan of this text")
                bounds = planned
    if bounds is None:
        bounds = ctx.shard_bounds(d_text, n, world)
    if sub_bounds is None:
        sub_bounds = bounds
    if text_rounds <= 0:
        text_rounds = 3 if index_bytes == 8 else 8
    nsh = world * k
    lo, hi = bounds[rank], bounds[rank + 1]
    dev = d_sa_full.device
    if d_grp is None and index_bytes != 4:
        raise ValueError("int64 rows run the wide engine, which always publishes its tie groups: pass d_grp")
    if d_grp is not None and d_grp.numel() < max(hi - lo, 1):
        raise ValueError(f"d_grp holds {d_grp.numel()} group heads, my slice has {hi - lo} rows")
    exchanging = _many(world) and gather_rows
    works = []
    unresolved_any, depth = False, 0

    def post(j):
        return allgatherv_ranges(d_sa_full, [(sub_bounds[g * k + j], sub_bounds[g * k + j + 1]) for g in range(world)], dist, wait=False)

    for j in range(k):
        sh = rank * k + j
        slo, shi = sub_bounds[sh], sub_bounds[sh + 1]
        reuse = j > 0
        if installs is not None and j > 0:          # (the first sub-shard's stripe sums were installed by plan_sharded)
            try:
                ctx.hist_install(sh, installs[j])
                reuse = False
            except _lib.MsufsortHipError:
                installs = None                      # local and recoverable: the remaining sub-shards count for themselves
                reuse = False
        elif installs is not None:
            reuse = False
        sl = d_sa_full[slo:shi] if shi > slo else torch.empty(1, dtype=d_sa_full.dtype, device=dev)
        if d_grp is None:
            ctx.make_sa_shard(d_text, n, sl, max(shi - slo, 1), sh, nsh, text_rounds=text_rounds, reuse_plan=reuse)
        else:
            gl = d_grp[slo - lo:shi - lo] if shi > slo else torch.empty(1, dtype=torch.int32, device=dev)
            _, _, unresolved, dep = ctx.make_sa_shard_groups(d_text, n, sl, gl, max(shi - slo, 1), sh, nsh, text_rounds=text_rounds,
                                                             index_bytes=index_bytes, verbose=verbose, reuse_plan=reuse)
            if shi > slo and slo > lo:
                gl += (slo - lo)                     # tie-group heads relative to MY slice (a tie group never spans two sub-shards)
            if unresolved:
                if unresolved_any and dep != depth:
                    raise _lib.MsufsortHipError(f"sub-shards stopped their key rounds at different depths ({depth}, {dep})")
                unresolved_any, depth = True, dep
        if j + 1 < k and exchanging:
            works += post(j)                         # travels while the next sub-shard is sorted
    if d_grp is None:
        if exchanging:
            works += post(k - 1)
            if overlap:
                return works
            wait_all(works, d_sa_full)
        return []
    # every unresolved rank stopped at the same depth (same number of rounds, same symbols per key): check it instead of
    # trusting it - doubling from a depth some group does not share would mis-sort silently
    big = 1 << 62
    flag = torch.tensor([depth if unresolved_any else 0, -(depth if unresolved_any else big)], dtype=torch.int64, device=dev)
    if _many(world):
        # (16 bytes, queued before the last sub-slice.  A complete build waits for the slices anyway, so the agreement travels on the same
        # communicator, in order behind the sub-slices already posted: no second communicator runs beside the exchange.  Only the
        # pipelined flavour, which returns without waiting, puts it on the small collectives' own communicator)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=hist_group if overlap else None)
        if gather_rows:
            works += post(k - 1)
    dmax, dmin = int(flag[0].item()), -int(flag[1].item())
    if dmax > 0:
        if dmin != dmax:
            raise _lib.MsufsortHipError(f"shards stopped their key rounds at different depths ({dmin} .. {dmax})")
        if _many(world) and not gather_rows:
            works = allgatherv_slices(d_sa_full, bounds, dist, wait=False)      # the doubling needs everybody's provisional rows
        wait_all(works, d_sa_full)
        if state is None:
            state = ShardState()
        _distributed_doubling(ctx, n, d_sa_full, d_grp, bounds, rank, world, dist, dmax, index_bytes, state, verbose)
        if _many(world) and gather_rows:
            allgatherv_slices(d_sa_full, bounds, dist)          # the final rows
        return []
    if overlap:
        return works
    wait_all(works, d_sa_full)
    return []


def plan_sub_bounds(ctx, d_text, n: int, world: int, k: int):
    """(bounds of the `world` rank slices, bounds of the world * k sub-shards) from ONE plan: the rank slices are unions of their
    sub-shards (the cuts of a world-shard plan need not be cuts of the finer plan: where a heavy two-byte key is refined, the
    tolerance follows the shard count)."""
    sub = ctx.shard_bounds(d_text, n, world * max(1, k))
    return list(sub[::max(1, k)]), list(sub)


def bwt_slice_bounds(bounds, sentinel_row: int):
    """Where every rank's BWT bytes go: row r of the suffix array contributes byte r - (r > sentinel_row) of the n-byte transform
    (the sentinel row - the row of suffix 0 - is removed, reference msufsort.cpp:1811-1815), so the slice of rows
    [bounds[g], bounds[g+1]) becomes the bytes [out[g], out[g+1])."""
    return [b - (1 if b > sentinel_row else 0) for b in bounds]


def forward_bwt_sharded(ctx, d_text, n: int, d_sa_full, bounds, rank: int, world: int, dist, d_bwt_out, d_row_bytes, index_bytes: int = 4, stats=None):
    """Forward BWT of a sharded build WITHOUT gathering the rows ("for BWT gather n/G-byte slices
    instead"; reference semantics msufsort.cpp:1771-1817: n bytes, sentinel row removed and returned).  My slice of d_sa_full
    must be final (build_sa_sharded(..., gather_rows=False)).  Every rank gathers the byte in front of each suffix of ITS rows
    (msufsort_hip_bwt_slice_dev), the ranks agree on the sentinel row (one all-reduce), every rank moves its bytes to their final
    positions in d_bwt_out (n bytes) and ONE all-gatherv of the byte slices completes the transform on every rank: n(G-1)/G bytes
    arrive per GPU instead of 4(n+1)(G-1)/G.  d_row_bytes: uint8 scratch of at least my slice's rows.  Returns the sentinel row."""
    import time

    import torch
    dev = d_bwt_out.device
    lo, hi = bounds[rank], bounds[rank + 1]
    s_local = -1
    if hi > lo:
        s_local = ctx.bwt_slice(d_text, n, d_sa_full[lo:hi], lo, hi, d_row_bytes, index_bytes)
    t0 = time.perf_counter()
    s = torch.tensor([s_local], dtype=torch.int64, device=dev)
    if _many(world):
        dist.all_reduce(s, op=dist.ReduceOp.MAX)
    sent = int(s.item())
    if not (1 <= sent <= n):
        raise _lib.MsufsortHipError(f"sharded forward BWT: no slice holds the row of suffix 0 (got {sent})")
    out = bwt_slice_bounds(bounds, sent)
    if hi > lo:
        if lo <= sent < hi:          # my slice holds the sentinel row: close the hole
            k = sent - lo
            if k:
                d_bwt_out[out[rank]:out[rank] + k] = d_row_bytes[:k]
            if hi - sent - 1 > 0:
                d_bwt_out[sent:sent + (hi - sent - 1)] = d_row_bytes[k + 1:hi - lo]
        else:
            d_bwt_out[out[rank]:out[rank + 1]] = d_row_bytes[:hi - lo]
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    if _many(world):
        allgatherv_slices(d_bwt_out, out, dist)
    if stats is not None:
        stats["bwt_exchange_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        stats["bwt_bytes_received"] = int(n - (out[rank + 1] - out[rank]))
    return sent
