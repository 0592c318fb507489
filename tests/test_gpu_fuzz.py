"""GPU fuzz parity: a few hundred structured inputs (random alphabets, runs, periodic blocks, trailing / embedded
zero runs, near-duplicate halves, sizes around the kernel size-class boundaries) against the oracle, bit-exact
for SA, BWT + sentinel, inverse BWT and LCP.  Deterministic seeds."""
import numpy as np
import pytest

from msufsort_amd import gen

pytestmark = pytest.mark.gpu


def _make(kind, n, seed):
    r = np.random.default_rng(seed)
    if kind == "alpha":                      # random alphabet size
        a = int(r.integers(1, 257))
        return r.integers(0, a, n, dtype=np.uint8)
    if kind == "runs":                       # long runs of equal bytes
        out = np.empty(n, np.uint8); p = 0
        while p < n:
            l = int(r.integers(1, 4000)); out[p:p + l] = r.integers(0, 4); p += l
        return out
    if kind == "periodic":                   # tandem blocks of random period
        out = np.empty(n, np.uint8); p = 0
        while p < n:
            per = int(r.integers(1, 300)); l = int(r.integers(per, 20 * per + 50))
            out[p:p + l] = np.resize(r.integers(97, 101, per, dtype=np.uint8), min(l, n - p)); p += l
        return out
    if kind == "zeros":                      # zero runs inside and at the end
        out = r.integers(0, 3, n, dtype=np.uint8)
        z = int(r.integers(0, max(1, n // 3))); out[n - z:] = 0
        q = int(r.integers(0, max(1, n // 2))); out[q:q + int(r.integers(0, 500))] = 0
        return out
    if kind == "dup":                        # second half = first half with a few edits (very long LCPs)
        h = r.integers(0, 256, n // 2, dtype=np.uint8)
        out = np.concatenate([h, h, r.integers(0, 256, n - 2 * (n // 2), dtype=np.uint8)])
        for _ in range(3):
            out[int(r.integers(0, n))] ^= 1
        return out
    if kind == "text":
        return gen.text_bytes(n, seed)
    raise ValueError(kind)


SIZES = [1, 2, 3, 31, 32, 33, 34, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1000, 4607, 4608, 4609, 9000, 18431, 18432, 18433,
         40000, 70001]


@pytest.mark.parametrize("kind", ["alpha", "runs", "periodic", "zeros", "dup", "text"])
def test_fuzz_kind(kind, oracle_mod):
    import msufsort_amd as M
    o = oracle_mod
    for i, n in enumerate(SIZES):
        for rep in range(2):
            t = np.ascontiguousarray(_make(kind, n, 1000 * i + rep + sum(kind.encode()) % 97))
            want = o.ref_make_suffix_array(t, 2) if (o.have_reference() and kind in ("periodic", "runs", "dup")) else o.make_suffix_array(t)
            sa = M.make_suffix_array(t)
            assert (sa == want).all(), (kind, n, rep)
            if rep == 0:
                b, s = M.forward_burrows_wheeler_transform(t)
                wb, ws = o.forward_bwt(t) if n < 20000 else (None, None)
                if wb is not None:
                    assert s == ws and (b == wb).all(), (kind, n)
                assert (M.reverse_burrows_wheeler_transform(b, s) == t).all(), (kind, n)
                assert (M.make_lcp_array(t, sa) == o.lcp(t, sa)).all(), (kind, n)


def test_fuzz_doubling_forced(oracle_mod):
    """Same inputs through the prefix-doubling path from the first round on."""
    import msufsort_amd as M
    o = oracle_mod
    for kind in ("periodic", "dup", "zeros"):
        for n in (513, 5000, 30000):
            t = np.ascontiguousarray(_make(kind, n, n))
            want = o.ref_make_suffix_array(t, 2) if o.have_reference() else o.make_suffix_array(t)
            assert (M.make_suffix_array(t, text_rounds=1) == want).all(), (kind, n)


@pytest.mark.parametrize("policy", ["1", "2"])
def test_fuzz_radix17_forced(oracle_mod, monkeypatch, policy):
    """The same inputs with the 17-bit front end forced at every size (policy 1: after the 16-bit histogram, 2: instead of it, with
    the fall-back when the 8-bit counters wrap): k_hist17 / k_scan17 / k_partition<512> / 512 children per segment on
    degenerate alphabets, runs, periods and near-duplicate halves - bit-exact SA through the one-shot entry point."""
    import msufsort_amd as M
    o = oracle_mod
    monkeypatch.setenv("MSUFSORT_HIP_RADIX17", policy)
    for kind in ("alpha", "runs", "periodic", "zeros", "dup", "text"):
        for i, n in enumerate((1, 2, 33, 513, 4609, 18433, 70001, 300000)):
            t = np.ascontiguousarray(_make(kind, n, 77 * i + sum(kind.encode()) % 89))
            want = o.ref_make_suffix_array(t, 2) if (o.have_reference() and kind in ("periodic", "runs", "dup")) else o.make_suffix_array(t)
            assert (M.make_suffix_array(t, two_stage=-1) == want).all(), (kind, n)


def test_fuzz_two_stage_sharded_first_stage(oracle_mod):
    """The two-stage build with its first stage cut into key-range shards (all of them on this GPU), forced on small structured
    inputs: whenever the call accepts the input (0) the rows are the reference's; periodic and duplicated inputs must be declined
    (1), never mis-sorted."""
    import torch
    import msufsort_amd as M
    o = oracle_mod
    ctx = M.DeviceContext(0)
    accepted = 0
    for kind in ("alpha", "runs", "periodic", "zeros", "dup", "text"):
        for i, n in enumerate((4096, 4609, 18433, 70001, 300000)):
            t = np.ascontiguousarray(_make(kind, n, 31 * i + sum(kind.encode()) % 83))
            d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
            sa = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
            bstar = torch.zeros(n // 2 + 2, dtype=torch.int32, device="cuda")
            for shards in (2, 5):
                r = ctx.make_sa_two_stage_sharded(d, n, sa, bstar, -1, shards, None, two_stage=1)
                assert r in (0, 1), (kind, n, shards, r)
                if r == 0:
                    want = o.ref_make_suffix_array(t, 2) if (o.have_reference() and kind in ("periodic", "runs", "dup")) else o.make_suffix_array(t)
                    assert (sa.cpu().numpy() == want).all(), (kind, n, shards)
                    accepted += 1
    assert accepted >= 10
