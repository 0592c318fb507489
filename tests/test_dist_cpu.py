"""CPU-only coverage of the N>1 host path: shard planning (host logic of the C-ABI) and the
all-gatherv of suffix-array slices over torch.distributed (gloo, world_size 2 and 3)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bstart(t):
    tt = np.concatenate([t, np.zeros(1, np.uint8)]).astype(np.uint32)
    h = np.bincount((tt[:-1] << 8) | tt[1:], minlength=65536)
    return np.concatenate([[0], np.cumsum(h)]).astype(np.uint64)


def test_plan_cuts_balanced_and_monotone():
    from msufsort_amd import dist as D
    from msufsort_amd import gen
    for name, n in (("random", 300000), ("text", 200000), ("dna", 100000)):
        t = gen.GENERATORS[name](n, 5)
        b = _bstart(t)
        for g in (1, 2, 3, 8):
            cuts, rows = D.plan_cuts(b, n, 0, g)
            assert cuts[0] == 0 and cuts[-1] == 65536 and rows[0] == 0 and rows[-1] == n + 1
            assert all(x <= y for x, y in zip(cuts, cuts[1:])) and all(x <= y for x, y in zip(rows, rows[1:]))
            for k in range(1, g):
                assert rows[k] == 1 + int(b[cuts[k]])
                # a cut never sits before the balanced target
                assert int(b[cuts[k]]) >= n * k // g
            if name == "random":
                sizes = np.diff(rows)
                assert sizes.max() - sizes.min() < 0.02 * n + 2
    # trailing zero run: its z rows and row 0 belong to shard 0
    cuts, rows = D.plan_cuts(_bstart(np.frombuffer(b"abcabcabc", np.uint8)), 12, 3, 2)
    assert rows[1] >= 1 + 3


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle
    from msufsort_amd import dist as D
    from msufsort_amd import gen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = gen.text_bytes(50000, 9)
        n = t.size
        want = oracle.make_suffix_array(t)              # the CHECKER; stands in for the per-rank HIP sort
        cuts, rows = D.plan_cuts(_bstart(t), n, 0, world)
        full = torch.full((n + 1,), -7, dtype=torch.int32)
        lo, hi = rows[rank], rows[rank + 1]
        full[lo:hi] = torch.from_numpy(want[lo:hi].copy())
        D.allgatherv_slices(full, rows, dist)
        ok = bool((full.numpy() == want).all())
        # slices really are key ranges: every suffix in slice g starts with a 16-bit key in [cuts[g], cuts[g+1])
        tt = np.concatenate([t, np.zeros(2, np.uint8)]).astype(np.uint32)
        keys = (tt[want[max(lo, 1):hi]] << 8) | tt[want[max(lo, 1):hi] + 1]
        ok = ok and bool(((keys >= cuts[rank]) & (keys < cuts[rank + 1])).all())
        open(os.path.join(tmp, f"r{rank}"), "w").write("ok" if ok else "bad")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgatherv_gloo(world, tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 1000 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"r{r}").read() == "ok"


def _worker_empty(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from msufsort_amd import dist as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # slice bounds as the sharded B* sort of a text produces them: ranks whose key range holds no B* suffix have EMPTY slices
        bounds = [0, 0, 1000, 1000, 2500][: world + 1] if world == 4 else [0, 700, 700, 1500]
        total = bounds[-1]
        full = torch.full((total,), -1, dtype=torch.int32)
        lo, hi = bounds[rank], bounds[rank + 1]
        full[lo:hi] = torch.arange(lo, hi, dtype=torch.int32) * 3 + 1
        D.allgatherv_slices(full, bounds, dist)
        ok = bool((full == torch.arange(total, dtype=torch.int32) * 3 + 1).all())
        open(os.path.join(tmp, f"e{rank}"), "w").write("ok" if ok else "bad")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 4])
def test_allgatherv_with_empty_slices_gloo(world, tmp_path):
    """The all-gatherv of the sorted-B* slices (msufsort_amd/dist.py::build_sa_two_stage_sharded) meets ranks without any B* suffix in
    their key range: empty slices must neither be sent nor waited for."""
    import torch.multiprocessing as mp
    port = 29700 + os.getpid() % 1000 + world
    mp.spawn(_worker_empty, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"e{r}").read() == "ok"


# ------------------------------------------------------------------------------------------------
# The multi-process driver of msufsort_amd/dist.py (sharded build -> distributed prefix doubling -> final gather, and the
# sharded forward BWT) with int32 AND int64 rows, on CPU tensors over gloo.  The per-shard device calls are stood in for by a
# small numpy model of the C-ABI pieces (same arguments, same update layout: ONE 64-bit word per update for int32 rows -
# new_row << 32 | suffix - TWO for int64 rows - {suffix, new_row}; include/msufsort_hip.h "Distributed prefix doubling").
# What this pins is the PYTHON side: slice-sized group buffers, window walking, update offsets (`pre`, e = 2), the windowed
# replica build, byte-slice bounds of the BWT.  The HIP kernels behind the same calls are covered by tests/test_gpu_dist.py.
# ------------------------------------------------------------------------------------------------
class _ModelEngine:
    def __init__(self, t, want, cuts, rows, depth):
        self.t, self.want, self.cuts, self.rows, self.depth = t, want, cuts, rows, depth
        self.pad = np.concatenate([t, np.zeros(depth + 8, np.uint8)])
        self.refine_ms = 0.0

    class _Tm:
        refine_ms = 0.0

    def timings(self):
        return self._Tm()

    def shard_bounds(self, d_text, n, world):
        return list(self.rows)

    def make_sa_shard_groups(self, d_text, n, sl, gl, capacity, shard, n_shards, *, verbose=0, text_rounds=0, index_bytes=4, reuse_plan=False):
        assert n_shards == len(self.rows) - 1
        self.calls = getattr(self, "calls", []) + [(shard, bool(reuse_plan))]
        lo, hi = self.rows[shard], self.rows[shard + 1]
        assert sl.dtype == (__import__("torch").int64 if index_bytes == 8 else __import__("torch").int32) and capacity >= hi - lo
        final = self.want[lo:hi].astype(np.int64)
        # provisional order: by the first `depth` bytes only (zero padded), ties in DESCENDING text position (anything but sorted)
        keys = np.stack([self.pad[np.minimum(final + k, len(self.pad) - 1)] for k in range(self.depth)], axis=1)
        keys[final == n] = 0
        order = np.lexsort([-final] + [keys[:, k] for k in range(self.depth - 1, -1, -1)])
        prov, pk = final[order], keys[order]
        new_grp = np.ones(hi - lo, bool)
        new_grp[1:] = (pk[1:] != pk[:-1]).any(axis=1)
        if lo == 0:
            new_grp[:2] = True           # row 0 (the empty suffix) is final by construction
        heads = np.maximum.accumulate(np.where(new_grp, np.arange(hi - lo), 0))
        sl.numpy()[:hi - lo] = prov
        gl.numpy()[:hi - lo] = heads
        unresolved = bool((np.bincount(heads) > 1).any())
        return lo, hi, unresolved, self.depth if unresolved else 0

    def isa_from_slice(self, sa_slice, grp_slice, lo, hi, isa, index_bytes=4):
        isa.numpy()[sa_slice.numpy()[:hi - lo]] = lo + grp_slice.numpy()[:hi - lo].astype(np.int64)

    def double_sort(self, n, sl, gl, gp, lo, hi, isa, h, index_bytes=4, verbose=0):
        sa, grp, prev, rank = sl.numpy(), gl.numpy(), gp.numpy(), isa.numpy()
        rows = hi - lo
        prev[:rows] = grp[:rows]
        sizes = np.bincount(grp[:rows], minlength=rows)
        tied = 0
        for head in np.nonzero(sizes > 1)[0]:
            tied += 1
            seg = sa[head:head + sizes[head]].astype(np.int64)
            key = np.where(seg + h <= n, rank[np.minimum(seg + h, n)], 0)
            o = np.argsort(key, kind="stable")
            seg, key = seg[o], key[o]
            sa[head:head + sizes[head]] = seg
            nh = np.ones(len(seg), bool)
            nh[1:] = key[1:] != key[:-1]
            grp[head:head + sizes[head]] = head + np.maximum.accumulate(np.where(nh, np.arange(len(seg)), 0))
        return tied, rows

    def emit_updates(self, sl, gl, gp, lo, hi, i0, i1, items_total, d_updates, capacity, index_bytes=4):
        sa, grp, prev, out = sl.numpy(), gl.numpy(), gp.numpy(), d_updates.numpy()
        rows = hi - lo
        r = np.arange(i0, i1)
        chg = r[grp[r] != prev[r]]
        assert len(chg) <= capacity
        if index_bytes == 8:
            out[0:2 * len(chg):2] = sa[chg]
            out[1:2 * len(chg):2] = lo + grp[chg].astype(np.int64)
        else:
            out[:len(chg)] = ((lo + grp[chg].astype(np.int64)) << 32) | sa[chg].astype(np.int64)
        sizes = np.bincount(grp[:rows], minlength=rows)
        return len(chg), int((sizes[grp[r]] > 1).sum())

    def apply_updates(self, d_updates, count, isa, index_bytes=4):
        u, rank = d_updates.numpy(), isa.numpy()
        if index_bytes == 8:
            rank[u[0:2 * count:2]] = u[1:2 * count:2]
        else:
            rank[u[:count] & 0xFFFFFFFF] = u[:count] >> 32

    # ---- the histogram computed sharded (dist.plan_sharded): 5 stripes of the text, the model's plan is self.rows ----
    STRIPES = 5

    def _stripe_counts(self, n, s, klo, khi):
        pos = np.arange(n * s // self.STRIPES, n * (s + 1) // self.STRIPES)
        key = (self.pad[pos].astype(np.int64) << 8) | self.pad[pos + 1]
        return np.bincount(self.pad[pos][(key >= klo) & (key < khi)], minlength=256)

    def hist_part(self, d_text, n, part, parts, d_hist):
        self.s0, self.s1 = self.STRIPES * part // parts, self.STRIPES * (part + 1) // parts
        pos = np.arange(n * self.s0 // self.STRIPES, n * self.s1 // self.STRIPES)
        d_hist.numpy()[:] = np.bincount((self.pad[pos].astype(np.int64) << 8) | self.pad[pos + 1], minlength=65536)
        return self.STRIPES, self.s0, self.s1

    def hist_plan(self, d_text, n, n_shards, d_hist_sum, d_sums):
        pos = np.arange(n)
        assert (d_hist_sum.numpy() == np.bincount((self.pad[pos].astype(np.int64) << 8) | self.pad[pos + 1], minlength=65536)).all()
        out = d_sums.numpy()
        out[:] = 0
        for g in range(n_shards):
            for s in range(self.s0, self.s1):
                out[g, s - self.s0] = self._stripe_counts(n, s, self.cuts[g], self.cuts[g + 1])
        self.n = n
        return list(self.rows)

    def hist_install(self, shard, d_stripe_sums):
        want = np.stack([self._stripe_counts(self.n, s, self.cuts[shard], self.cuts[shard + 1]) for s in range(self.STRIPES)])
        assert d_stripe_sums.shape == (self.STRIPES, 256) and (d_stripe_sums.numpy() == want).all()
        self.installed = getattr(self, "installed", 0) + 1

    def bwt_slice(self, d_text, n, d_sa_slice, lo, hi, d_row_bytes, index_bytes=4):
        sa = d_sa_slice.numpy()[:hi - lo].astype(np.int64)
        d_row_bytes.numpy()[:hi - lo] = np.where(sa > 0, self.t[np.maximum(sa - 1, 0)], 0)
        z = np.nonzero(sa == 0)[0]
        return int(lo + z[0]) if len(z) else -1


def _worker_doubling(rank, world, port, tmp, index_bytes, k=1):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MSUFSORT_DIST_WINDOW="700",          # several update / group-head windows per step on a small input
                      MSUFSORT_DIST_SHARDED_HIST="1")      # ... and the histogram counted 1/world per rank at any size
    import torch
    import torch.distributed as dist

    import oracle
    from msufsort_amd import dist as D
    from msufsort_amd import gen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = gen.dna_tandem_bytes(24000, 9)
        n = t.size
        want = oracle.make_suffix_array(t)
        wb, ws = oracle.forward_bwt(t)
        # k sub-shards per rank (dist.sub_shards_for): the model's plan is the plan of world * k shards, a rank's slice their union
        cuts, sub_rows = D.plan_cuts(_bstart(t), n, 0, world * k)
        eng = _ModelEngine(t, want, cuts, sub_rows, depth=6)
        rows = sub_rows[::k]
        sub_arg = sub_rows if k > 1 else None
        dt = torch.int64 if index_bytes == 8 else torch.int32
        full = torch.full((n + 1,), -7, dtype=dt)
        rows_max = max(rows[g + 1] - rows[g] for g in range(world))
        d_grp = torch.zeros(rows_max, dtype=torch.int32)          # slice sized: NOT n + 1
        state = D.ShardState()
        hst = {}
        D.build_sa_sharded(eng, None, n, full, rank, world, dist, rows, d_grp=d_grp, index_bytes=index_bytes, state=state, stats=hst, sub_bounds=sub_arg)
        ok = hst.get("sharded_hist") == 1 and eng.installed == k      # the histogram was counted 1/world per rank and the stripe sums of each of my sub-shards arrived
        ok = ok and eng.calls == [(rank * k + j, False) for j in range(k)]      # (installed stripe sums: no sub-shard reuses a plan)
        ok = ok and bool((full.numpy() == want).all()) and state.stats["doubling_steps"] >= 2 and state.stats["updates"] > 0
        ok = ok and state.stats["windows"] > state.stats["doubling_steps"] and state.stats["index_bytes"] == index_bytes
        # rows kept distributed + the sharded forward transform: n bytes exchanged instead of the rows
        full2 = torch.full((n + 1,), -7, dtype=dt)
        eng.calls = []
        os.environ["MSUFSORT_DIST_SHARDED_HIST"] = "0"          # replicated histogram: the first sub-shard plans, the others reuse its plan
        D.build_sa_sharded(eng, None, n, full2, rank, world, dist, rows, d_grp=d_grp, index_bytes=index_bytes, state=state, gather_rows=False, sub_bounds=sub_arg)
        ok = ok and eng.calls == [(rank * k + j, j > 0) for j in range(k)]
        lo, hi = rows[rank], rows[rank + 1]
        ok = ok and bool((full2.numpy()[lo:hi] == want[lo:hi]).all())
        bwt = torch.zeros(n, dtype=torch.uint8)
        st = {}
        sent = D.forward_bwt_sharded(eng, None, n, full2, rows, rank, world, dist, bwt, torch.zeros(rows_max, dtype=torch.uint8), index_bytes, stats=st)
        ok = ok and sent == ws and bool((bwt.numpy() == wb).all()) and st["bwt_bytes_received"] == n - (D.bwt_slice_bounds(rows, ws)[rank + 1] - D.bwt_slice_bounds(rows, ws)[rank])
        open(os.path.join(tmp, f"d{rank}"), "w").write("ok" if ok else f"bad {state.stats}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,index_bytes,k", [(2, 4, 1), (2, 8, 2), (3, 8, 1), (4, 4, 3)])
def test_distributed_doubling_driver_gloo(world, index_bytes, k, tmp_path, oracle_mod):
    """dist.build_sa_sharded -> _distributed_doubling -> forward_bwt_sharded with int32 and int64 rows (BASELINE config 5 is the
    int64 flavour over 8 ranks): result == the oracle's suffix array and BWT on every rank.  k > 1: every rank sorts its key range as
    k sub-shards and posts each sub-slice while the next one is sorted (round 6) - same rows, group heads moved to the rank's slice."""
    import torch.multiprocessing as mp
    port = 29900 + os.getpid() % 1000 + 8 * world + index_bytes
    mp.spawn(_worker_doubling, args=(world, port, str(tmp_path), index_bytes, k), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"d{r}").read() == "ok"


def test_sharded_histogram_policy(monkeypatch):
    """dist.sharded_hist_enabled: on where >= 0.75 GiB of text are counted by the OTHER ranks (measured break-even);
    MSUFSORT_DIST_SHARDED_HIST=1 / 0 force it; one rank only under the always-collective hook."""
    from msufsort_amd import dist as D
    monkeypatch.delenv("MSUFSORT_DIST_SHARDED_HIST", raising=False)
    monkeypatch.delenv("MSUFSORT_DIST_ALWAYS_COLLECTIVE", raising=False)
    n1 = (1 << 30) - 1
    assert D.sharded_hist_enabled(8, n1) and not D.sharded_hist_enabled(4, n1) and not D.sharded_hist_enabled(2, n1)
    assert D.sharded_hist_enabled(4, 1 << 30) and D.sharded_hist_enabled(2, 3 << 29) and not D.sharded_hist_enabled(8, 1 << 28)
    assert D.sharded_hist_enabled(8) and not D.sharded_hist_enabled(1, 1 << 33)
    monkeypatch.setenv("MSUFSORT_DIST_SHARDED_HIST", "1")
    assert D.sharded_hist_enabled(2, 1000) and not D.sharded_hist_enabled(1, 1 << 33)
    monkeypatch.setenv("MSUFSORT_DIST_ALWAYS_COLLECTIVE", "1")
    assert D.sharded_hist_enabled(1, 1000)
    monkeypatch.setenv("MSUFSORT_DIST_SHARDED_HIST", "0")
    assert not D.sharded_hist_enabled(8, 1 << 33)


def test_update_offsets_and_bwt_bounds():
    """The two pieces of index arithmetic the ranks must agree on: word offsets of a window's updates (one word per update for
    int32 rows, two for int64 rows) and the byte ranges of the BWT slices around the removed sentinel row."""
    from msufsort_amd import dist as D
    assert D.update_offsets([3, 0, 5], 4) == ([0, 3, 3, 8], 1)
    assert D.update_offsets([3, 0, 5], 8) == ([0, 6, 6, 16], 2)
    assert D.update_offsets([], 8) == ([0], 2)
    # rows 0..10 (n = 10), sentinel row 4 inside the second slice: bytes 0..2 | 3..5 (row 4 dropped) | 6..9
    assert D.bwt_slice_bounds([0, 3, 7, 11], 4) == [0, 3, 6, 10]
    assert D.bwt_slice_bounds([0, 3, 7, 11], 3) == [0, 3, 6, 10]          # sentinel = first row of a slice
    assert D.bwt_slice_bounds([0, 3, 7, 11], 10) == [0, 3, 7, 10]         # ... = last row of all
    assert D.ShardState.bytes_needed(1 << 33, 1 << 30, 8, 8) < 80 << 30   # 64 GiB replica + windows


def _worker_hist_failure(rank, world, port, tmp, stage):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MSUFSORT_DIST_SHARDED_HIST="1")
    import datetime

    import torch
    import torch.distributed as dist

    import oracle
    from msufsort_amd import _lib
    from msufsort_amd import dist as D
    from msufsort_amd import gen
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        t = gen.random_bytes(20000, 5)
        n = t.size
        cuts, rows = D.plan_cuts(_bstart(t), n, 0, world)
        eng = _ModelEngine(t, oracle.make_suffix_array(t), cuts, rows, depth=6)
        if rank == 1:          # ONE rank fails, in the C call of the given stage
            def boom(*a, **k):
                raise _lib.MsufsortHipError("injected failure in " + stage)
            setattr(eng, stage, boom)
        out = "returned"
        try:
            got = D.plan_sharded(eng, torch.zeros(1, dtype=torch.uint8), n, rank, world, dist)
            out = "bounds" if got == list(rows) else f"other {got}"
        except _lib.MsufsortHipError as e:
            out = "raised: " + str(e)
        open(os.path.join(tmp, f"h{rank}"), "w").write(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("stage", ["hist_part", "hist_plan", "hist_install"])
def test_sharded_histogram_failure_on_one_rank_reaches_every_rank(stage, tmp_path, oracle_mod):
    """dist.plan_sharded: a rank-local failure between its collectives must not leave the peers blocked in the next one (round-5
    advisor finding).  hist_part / hist_plan: every rank raises (the status travels with the totals / ahead of the all-gather);
    hist_install comes after the last collective: the failing rank alone falls back to counting for itself - same plan."""
    import torch.multiprocessing as mp
    world = 3
    port = 29300 + os.getpid() % 500 + {"hist_part": 0, "hist_plan": 1, "hist_install": 2}[stage]
    mp.spawn(_worker_hist_failure, args=(world, port, str(tmp_path), stage), nprocs=world, join=True)
    res = [open(tmp_path / f"h{r}").read() for r in range(world)]
    if stage == "hist_install":
        assert res == ["bounds"] * world, res
    else:
        assert all(r.startswith("raised") for r in res), res
        assert "injected failure" in res[1] and "a peer failed" in res[0] and "a peer failed" in res[2], res
