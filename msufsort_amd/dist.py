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


class ShardState:
    """Per-rank buffers of the distributed prefix doubling (allocated on first use, reused by later builds)."""

    def __init__(self):
        self.isa = None
        self.grp_prev = None
        self.upd_local = None
        self.upd_all = None
        self.stats = {}

    def ensure(self, n, rows_max, world, index_bytes, device):
        import torch
        dt = torch.int64 if index_bytes == 8 else torch.int32
        if self.isa is None or self.isa.numel() < n + 2 or self.isa.dtype != dt:
            self.isa = torch.empty(n + 2, dtype=dt, device=device)
        if self.grp_prev is None or self.grp_prev.numel() < rows_max:
            self.grp_prev = torch.empty(max(rows_max, 1), dtype=torch.int32, device=device)
        e = 2 if index_bytes == 8 else 1
        self.win = max(1, min(rows_max, 1 << 25))
        if self.upd_local is None or self.upd_local.numel() < self.win * e:
            self.upd_local = torch.empty(self.win * e, dtype=torch.int64, device=device)
        if self.upd_all is None or self.upd_all.numel() < self.win * e * world:
            self.upd_all = torch.empty(self.win * e * world, dtype=torch.int64, device=device)


def _distributed_doubling(ctx, n, d_sa_full, d_grp_full, bounds, rank, world, dist, depth, index_bytes, state, verbose=0):
    """Prefix doubling over the shards (include/msufsort_hip.h, 'Distributed prefix doubling'): every rank sorts only the
    tie groups of its own slice; the rank array is replicated and refreshed once per step with ONE all-gatherv of the
    (suffix, new head row) updates of all ranks.  Starts from gathered provisional rows + group heads."""
    import time

    import torch
    dev = d_sa_full.device
    lo, hi = bounds[rank], bounds[rank + 1]
    rows_max = max(bounds[g + 1] - bounds[g] for g in range(world))
    state.ensure(n, rows_max, world, index_bytes, dev)
    isa, e = state.isa, (2 if index_bytes == 8 else 1)
    one = torch.empty(1, dtype=d_sa_full.dtype, device=dev)
    sl = d_sa_full[lo:hi] if hi > lo else one
    gl = d_grp_full[lo:hi] if hi > lo else torch.empty(1, dtype=torch.int32, device=dev)
    gp = state.grp_prev
    for g in range(world):
        if bounds[g + 1] > bounds[g]:
            ctx.isa_from_slice(d_sa_full[bounds[g]:bounds[g + 1]], d_grp_full[bounds[g]:bounds[g + 1]], bounds[g], bounds[g + 1], isa, index_bytes)
    st = {"doubling_steps": 0, "sort_ms": 0.0, "exchange_ms": 0.0, "updates": 0, "depth": depth}
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
        if world > 1:
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
            if world > 1:
                dist.all_reduce(counts)
            cl = [int(x) for x in counts.tolist()]
            pre = [0]
            for x in cl:
                pre.append(pre[-1] + x * e)
            if pre[-1]:
                if cnt:
                    state.upd_all[pre[rank]:pre[rank + 1]] = state.upd_local[:cnt * e]
                if dev.type == "cuda":
                    torch.cuda.current_stream(dev).synchronize()
                if world > 1:
                    allgatherv_slices(state.upd_all, pre, dist)
                st["exchange_ms"] += (time.perf_counter() - t0) * 1e3
                ctx.apply_updates(state.upd_all, pre[-1] // e, isa, index_bytes)
                st["updates"] += pre[-1] // e
        tt = torch.tensor([tied_local], dtype=torch.int64, device=dev)
        if world > 1:
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
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        agreed = int(flag.item())
        if agreed == 0 and world > 1:
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


def build_sa_sharded(ctx, d_text, n: int, d_sa_full, rank: int, world: int, dist, bounds=None, text_rounds: int = 8,
                     d_grp_full=None, overlap=False, index_bytes: int = 4, state=None, verbose: int = 0):
    """One step of the sharded build on this rank: sort my key range into my slice, all-gatherv the slices.

    Deep ties (long repeats) cannot be finished by key gathers.  With `d_grp_full` (int32 view of uint32, n+1) every rank
    also publishes the tie groups of its slice; if any rank stopped with unresolved groups the provisional rows and the
    group heads are gathered once, every rank builds its replica of the rank array, and the ranks run the DISTRIBUTED
    prefix doubling: each sorts only its own groups, one all-gatherv of rank updates per step.  The final slices are
    gathered at the end.  Without `d_grp_full` such inputs raise (MSUFSORT_HIP_ERR_UNSUPPORTED).
    index_bytes = 8: wide engine (int64 rows; any n up to 2^40 - 2).

    overlap=True: returns the pending exchange handles instead of waiting, so the caller can start the next build
    (into ANOTHER output buffer) while the slices travel; finish with `wait_all(works, d_sa_full)`.  If the build
    turns out to need the doubling phase everything is completed here and [] is returned."""
    import torch
    if bounds is None:
        bounds = ctx.shard_bounds(d_text, n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    dev = d_sa_full.device
    sl = d_sa_full[lo:hi] if hi > lo else torch.empty(1, dtype=d_sa_full.dtype, device=dev)
    if d_grp_full is None:
        assert index_bytes == 4
        ctx.make_sa_shard(d_text, n, sl, max(hi - lo, 1), rank, world, text_rounds=text_rounds)
        if world > 1:
            works = allgatherv_slices(d_sa_full, bounds, dist, wait=not overlap)
            return works if overlap else []
        return []
    gl = d_grp_full[lo:hi] if hi > lo else torch.empty(1, dtype=torch.int32, device=dev)
    _, _, unresolved, depth = ctx.make_sa_shard_groups(d_text, n, sl, gl, max(hi - lo, 1), rank, world, text_rounds=text_rounds,
                                                       index_bytes=index_bytes, verbose=verbose)
    # every unresolved rank stopped at the same depth (same number of rounds, same symbols per key): check it instead of
    # trusting it - doubling from a depth some group does not share would mis-sort silently
    big = 1 << 62
    flag = torch.tensor([depth if unresolved else 0, -(depth if unresolved else big)], dtype=torch.int64, device=dev)
    works = []
    if world > 1:
        w = dist.all_reduce(flag, op=dist.ReduceOp.MAX, async_op=True)       # rides along with the slice exchange
        works = allgatherv_slices(d_sa_full, bounds, dist, wait=False)
        w.wait()
    dmax, dmin = int(flag[0].item()), -int(flag[1].item())
    if dmax > 0:
        if dmin != dmax:
            raise _lib.MsufsortHipError(f"shards stopped their key rounds at different depths ({dmin} .. {dmax})")
        wait_all(works, d_sa_full)
        if world > 1:
            allgatherv_slices(d_grp_full, bounds, dist)
        if state is None:
            state = ShardState()
        _distributed_doubling(ctx, n, d_sa_full, d_grp_full, bounds, rank, world, dist, dmax, index_bytes, state, verbose)
        if world > 1:
            allgatherv_slices(d_sa_full, bounds, dist)          # the final rows
        return []
    if overlap:
        return works
    wait_all(works, d_sa_full)
    return []
