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
    """Host-only: (cuts, rows) for `n_shards` shards from the exclusive 16-bit-key prefix bstart[65537]."""
    b = np.ascontiguousarray(bstart, dtype=np.uint32)
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
    ok = 1
    try:
        probe = torch.zeros(world * 4, dtype=torch.int32, device=device)
        probe[rank * 4:(rank + 1) * 4] = rank + 1
        _MODE["mode"] = "p2p"
        allgatherv_slices(probe, [4 * g for g in range(world + 1)], dist)
        if not bool((probe.view(world, 4) == torch.arange(1, world + 1, device=device, dtype=torch.int32)[:, None]).all()):
            ok = 0
    except Exception:  # noqa: BLE001
        ok = 0
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


def build_sa_sharded(ctx, d_text, n: int, d_sa_full, rank: int, world: int, dist, bounds=None, text_rounds: int = 8,
                     d_grp_full=None, overlap=False):
    """One step of the sharded build on this rank: sort my key range into my slice, all-gatherv the slices.

    Deep ties (long repeats) cannot be finished shard-locally - prefix doubling needs the ranks of ALL suffixes.
    With `d_grp_full` (int32, n+1) every rank also publishes the tie groups of its slice; if any rank stopped with
    unresolved groups the group slices are gathered too and every rank finishes the complete array by prefix
    doubling (replicated).  Without it such inputs raise (MSUFSORT_HIP_ERR_UNSUPPORTED).

    overlap=True: returns the pending exchange handles instead of waiting, so the caller can start the next build
    (into ANOTHER output buffer) while the slices travel; finish with `wait_all(works, d_sa_full)`.  If the build
    turns out to need the finishing pass the exchange is completed here and [] is returned."""
    import torch
    if bounds is None:
        bounds = ctx.shard_bounds(d_text, n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    dev = d_sa_full.device
    sl = d_sa_full[lo:hi] if hi > lo else torch.empty(1, dtype=torch.int32, device=dev)
    if d_grp_full is None:
        ctx.make_sa_shard(d_text, n, sl, max(hi - lo, 1), rank, world, text_rounds=text_rounds)
        if world > 1:
            works = allgatherv_slices(d_sa_full, bounds, dist, wait=not overlap)
            return works if overlap else []
        return []
    gl = d_grp_full[lo:hi] if hi > lo else torch.empty(1, dtype=torch.int32, device=dev)
    _, _, unresolved, depth = ctx.make_sa_shard_groups(d_text, n, sl, gl, max(hi - lo, 1), rank, world, text_rounds=text_rounds)
    flag = torch.tensor([depth if unresolved else 0], dtype=torch.int64, device=dev)
    works = []
    if world > 1:
        w = dist.all_reduce(flag, op=dist.ReduceOp.MAX, async_op=True)       # rides along with the slice exchange
        works = allgatherv_slices(d_sa_full, bounds, dist, wait=False)
        w.wait()
    depth = int(flag.item())
    if depth > 0:
        wait_all(works, d_sa_full)
        if world > 1:
            allgatherv_slices(d_grp_full, bounds, dist)
        ctx.finish_sa(d_text, n, d_sa_full, d_grp_full, depth)
        return []
    if overlap:
        return works
    wait_all(works, d_sa_full)
    return []
