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
