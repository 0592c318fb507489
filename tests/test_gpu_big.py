"""BASELINE config 5 territory on ONE MI355X: inputs beyond 2^31 - 2 bytes through the wide engine (40-bit indices,
int64 rows) with logical shards taking turns on the GPU and the distributed prefix doubling, checked by the 64-bit
on-device checker (exact: permutation + linear-time rank order test).  The same engine is bit-exact against the
reference on small inputs (test_gpu_parity.py::test_wide_engine_parity) and on the first 2^30 - 1 bytes of the DNA
stream (golden hashes, test_full_size_golden)."""
import os
import time

import numpy as np
import pytest

from msufsort_amd import gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dna_gpu(n, seed, dev):
    """gen.dna_bytes(n, seed) computed on the GPU (same splitmix64 stream, same bytes), in blocks."""
    import torch
    out = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    out[n:] = 0
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    blk = 1 << 25                                    # draws per block (8 bytes each)
    nd = (n + 7) // 8
    for s in range(0, nd, blk):
        c = min(blk, nd - s)
        k = torch.arange(s + 1, s + c + 1, dtype=torch.int64, device=dev)
        z = (k * np.int64(np.uint64(0x9E3779B97F4A7C15).astype(np.int64))) + np.int64(seed)
        def shr(x, b):                                # logical shift right on int64
            return (x >> b) & ((1 << (64 - b)) - 1)
        z = (z ^ shr(z, 30)) * np.int64(np.uint64(0xBF58476D1CE4E5B9).astype(np.int64))
        z = (z ^ shr(z, 27)) * np.int64(np.uint64(0x94D049BB133111EB).astype(np.int64))
        z = z ^ shr(z, 31)
        b = z.view(torch.uint8)                       # little-endian bytes of every draw
        take = min(b.numel(), n - s * 8)
        out[s * 8: s * 8 + take] = lut[(b[:take] & 3).long()]
    return out


def test_dna_generator_on_gpu_matches_host():
    import torch
    n = 100003
    d = _dna_gpu(n, 7, torch.device("cuda"))
    assert (d[:n].cpu().numpy() == gen.dna_bytes(n, 7)).all()


@pytest.mark.parametrize("n,shards,need_gb", [((1 << 32) + 12345, 8, 200), (1 << 33, 32, 255)])
def test_beyond_int32_logical_shards(n, shards, need_gb):
    """n = 2^32 + 12,345, and BASELINE config 5 at its full size (8 GiB of DNA; there the 8 GPUs' shards become 32 logical
    shards that take turns on the one GPU of the box - ~230 GB of its 288 GB of HBM)."""
    import torch

    import msufsort_amd as M
    dev = torch.device("cuda")
    free, total = torch.cuda.mem_get_info()
    if free < need_gb << 30:
        pytest.skip(f"needs ~{need_gb} GB of free HBM")
    t0 = time.time()
    d = _dna_gpu(n, 2024, dev)
    # long repeats: 64 MiB of tandem-repeat DNA spliced in, and one 1 MiB block copied 3 GiB further on
    tr = gen.dna_tandem_bytes(1 << 26, 11)
    d[1 << 30: (1 << 30) + (1 << 26)] = torch.from_numpy(tr).to(dev)
    d[(7 << 29): (7 << 29) + (1 << 20)] = d[12345: 12345 + (1 << 20)].clone()
    if n > (6 << 30):
        d[(11 << 29): (11 << 29) + (1 << 26)] = torch.from_numpy(gen.dna_tandem_bytes(1 << 26, 12)).to(dev)
    torch.cuda.synchronize()
    t1 = time.time()
    ctx = M.DeviceContext(0)
    sa = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ctx.make_sa_i64(d, n, sa, n_shards=shards, verbose=int(os.environ.get("MSUFSORT_TEST_VERBOSE", "0")))
    t2 = time.time()
    sa.fill_(-1)
    ctx.make_sa_i64(d, n, sa, n_shards=shards)       # again, with every buffer in place (the first call allocates ~200 GB)
    t2b = time.time()
    tm = ctx.timings()
    assert tm.logical_shards >= shards and tm.doubling_rounds >= 1
    assert int(sa[0]) == n
    ctx.trim()
    errs = ctx.validate_sa(d, n, sa, index_bytes=8)
    t3 = time.time()
    print(f"\nn={n}: generate {t1 - t0:.1f}s, wide SA build {t2 - t1:.2f}s with allocations, {t2b - t2:.2f}s again ({tm.logical_shards} logical shards, depth {tm.stop_depth}, "
          f"{tm.doubling_rounds} doubling steps, doubling {tm.other_ms:.0f} ms), check {t3 - t2b:.1f}s, errors {errs}")
    assert errs == 0
    # the checker sees damage at this size too
    bad = sa[: 1 << 20].clone()
    sa[5], sa[6] = int(bad[6]), int(bad[5])
    assert ctx.validate_sa(d, n, sa, index_bytes=8) > 0
    del sa, d, bad
    ctx.trim()
    torch.cuda.empty_cache()


def test_config5_own_stream_full_size():
    """BASELINE config 5 on ITS OWN stream: alternating random stretches and tandem blocks (unit 1-50, 10-2010
    copies) end to end - gen.dna_tandem_bytes, the generator of the 2^28 golden vector - at the full n = 2^33, int64 rows,
    32 logical shards on the one GPU, distributed prefix doubling; 64-bit on-device checker (order + permutation)."""
    import torch

    import msufsort_amd as M
    dev = torch.device("cuda")
    free, total = torch.cuda.mem_get_info()
    if free < 255 << 30:
        pytest.skip("needs ~255 GB of free HBM")
    n = 1 << 33
    t0 = time.time()
    t = gen.dna_tandem_bytes(n, 9)
    # (the first 2^28 bytes are the golden input whose reference hashes tests/golden/golden_full.json holds)
    import json
    g = [e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json")))["full"] if e["generator"] == "dna_tandem"][0]
    import oracle
    assert "%016x" % oracle.fnv1a64(t[: 1 << 28]) == g["input_fnv"]
    d = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    for s in range(0, n, 1 << 30):
        d[s: s + (1 << 30)] = torch.from_numpy(t[s: s + (1 << 30)]).to(dev)
    del t
    torch.cuda.synchronize()
    t1 = time.time()
    ctx = M.DeviceContext(0)
    sa = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ctx.make_sa_i64(d, n, sa, n_shards=32)
    t2 = time.time()
    tm = ctx.timings()
    assert int(sa[0]) == n and tm.logical_shards >= 32 and tm.doubling_rounds >= 1
    ctx.trim()
    errs = ctx.validate_sa(d, n, sa, index_bytes=8)
    t3 = time.time()
    print(f"\nconfig 5 stream, n=2^33: generate + upload {t1 - t0:.1f}s, wide SA build {t2 - t1:.2f}s with allocations ({tm.logical_shards} logical shards, "
          f"depth {tm.stop_depth}, {tm.doubling_rounds} doubling steps, doubling {tm.other_ms:.0f} ms), check {t3 - t2:.1f}s, errors {errs}")
    assert errs == 0
    del sa, d
    ctx.trim()
    torch.cuda.empty_cache()


def _random_gpu(n, seed, dev):
    """gen.random_bytes(n, seed) computed on the GPU (same splitmix64 stream)."""
    import torch
    out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    blk = 1 << 25
    nd = (n + 7) // 8
    c1, c2, c3 = (np.int64(np.uint64(x).astype(np.int64)) for x in (0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB))
    for s in range(0, nd, blk):
        c = min(blk, nd - s)
        k = torch.arange(s + 1, s + c + 1, dtype=torch.int64, device=dev)
        z = k * c1 + np.int64(seed)
        shr = lambda x, b: (x >> b) & ((1 << (64 - b)) - 1)   # noqa: E731
        z = (z ^ shr(z, 30)) * c2
        z = (z ^ shr(z, 27)) * c3
        z = z ^ shr(z, 31)
        b = z.view(torch.uint8)
        take = min(b.numel(), n - s * 8)
        out[s * 8: s * 8 + take] = b[:take]
    return out


@pytest.mark.parametrize("mib", [285, 300, 540, 1090, 1122, 1180, 1300, 2047])
def test_bucket_sort_at_the_class_limits(mib):
    """Random bytes whose 65,536 two-byte buckets sit at the limits of the LDS sorts: ~4560 records (fills the 256-thread
    shape of k_sort_bits, limit 4608), ~4800 (just over: the 1024-thread shape a quarter full), ~17,440 and ~17,950 (fill the
    1024-thread shape, limit 18,432: its dirty list is nearly full), ~18,880, ~20,800 and ~32,750 (over: since round 4 the level-1
    partition splits 512 ways on a 17-bit histogram, k_hist17 + k_partition<512>, instead of a third partition level; 300 and
    540 MiB take the same 17-bit path so that their children fill the 4608-record shape).  Rows checked on the device; all but
    a handful of segments must stay with k_sort_bits at every size, and the level structure is the expected one."""
    import torch

    import msufsort_amd as M
    dev = torch.device("cuda")
    n = (mib << 20) + 7919
    d = _random_gpu(n, 1000 + mib, dev)
    ctx = M.DeviceContext(0)
    sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ctx.make_sa(d, n, sa)
    ctx.make_sa(d, n, sa)            # (the first build of a process also loads the code objects: time the second)
    tm = ctx.timings()
    assert int(sa[0]) == n and ctx.validate_sa(d, n, sa) == 0
    assert tm.bucket_sort_handed_back < 512, tm.bucket_sort_handed_back     # (of 65,536 / 131,072: a few with an overfull dirty list are
                                                                              # expected, thousands were the cliff next to the limits)
    assert tm.radix_bits == (17 if mib in (300, 540, 1180, 1300, 2047) else 16), tm.radix_bits
    assert tm.total_ms < 13.0 * n / 2**30, tm.total_ms                      # no size pays 2x per byte any more (round 3: 17 ms per GiB at 1180 MiB)
    del sa, d
    ctx.trim(); torch.cuda.empty_cache()


def test_hist17_first_declines_text():
    """At sizes where uniform bytes would take the 17-bit levels (two-byte buckets of 4.6 - 8.9 K or above 18.3 K on average) the 17-bit
    histogram runs first.  A text of such a size makes its 8-bit counters wrap: the build must fall back to the 16-bit histogram
    and levels (sort-all path forced here; rows checked on the device)."""
    import torch

    import msufsort_amd as M
    n = (300 << 20) + 123
    t = gen.text_bytes(n, 41)
    d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
    ctx = M.DeviceContext(0)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, sa, two_stage=-1)
    assert ctx.timings().radix_bits == 16 and ctx.validate_sa(d, n, sa) == 0
    del sa, d
    ctx.trim(); torch.cuda.empty_cache()


def test_int32_limit_and_first_wide_size():
    """The two sides of the index-width boundary on random bytes: n = 2^31 - 2 (largest int32 build, narrow engine) and
    n = 2^31 + 5 through the int64 entry point (smallest input that MUST take the wide engine), both checked on device."""
    import torch

    import msufsort_amd as M
    dev = torch.device("cuda")
    free, _ = torch.cuda.mem_get_info()
    if free < 230 << 30:
        pytest.skip("needs ~230 GB of free HBM")
    n = (1 << 31) - 2
    d = _random_gpu((1 << 31) + 5, 77, dev)
    assert (d[:100003].cpu().numpy() == gen.random_bytes(100003, 77)).all()
    ctx = M.DeviceContext(0)
    sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    keep = d[n:n + 64].clone()
    d[n:n + 64] = 0                                  # the int32 build sees the first 2^31 - 2 bytes (+ its zero pad)
    t0 = time.time()
    ctx.make_sa(d, n, sa)
    t1 = time.time()
    assert int(sa[0]) == n and ctx.validate_sa(d, n, sa) == 0
    print(f"\nn=2^31-2 random, int32 rows: {t1 - t0:.3f}s")
    del sa
    ctx.trim(); torch.cuda.empty_cache()
    d[n:n + 64] = keep
    n = (1 << 31) + 5
    sa64 = torch.empty(n + 1, dtype=torch.int64, device=dev)
    t0 = time.time()
    ctx.make_sa_i64(d, n, sa64)
    t1 = time.time()
    tm = ctx.timings()
    assert tm.logical_shards >= 2                       # logical shards: the wide engine ran
    ctx.trim()
    assert int(sa64[0]) == n and ctx.validate_sa(d, n, sa64, index_bytes=8) == 0
    print(f"n=2^31+5 random, int64 rows, wide engine ({tm.logical_shards} logical shards): {t1 - t0:.3f}s")
    # forward BWT beyond the int32 rows (wide engine inside): every byte equal to T[SA[r] - 1], sentinel row where SA[r] == 0
    bwt = torch.empty(n, dtype=torch.uint8, device=dev)
    sent = ctx.forward_bwt(d, n, bwt)
    assert int(sa64[sent]) == 0
    rows = torch.cat([sa64[:sent], sa64[sent + 1:]])
    assert torch.equal(bwt, d[rows - 1])
    del bwt, rows
    del sa64, d
    ctx.trim(); torch.cuda.empty_cache()


def test_two_stage_at_int32_limit():
    """The two-stage build (B* sort + induction) on the largest input the int32 rows allow: 2^31 - 2 bytes over a 30-letter
    alphabet; rows checked on device, forward BWT from the rows' preceding characters against the gathered one."""
    import torch

    import msufsort_amd as M
    dev = torch.device("cuda")
    free, _ = torch.cuda.mem_get_info()
    if free < 150 << 30:
        pytest.skip("needs ~150 GB of free HBM")
    n = (1 << 31) - 2
    d = _random_gpu(n, 99, dev)
    lut = torch.tensor(list(b"etaoinshrdlucmfwypvbgkjqxz  \n.,e"), dtype=torch.uint8, device=dev)
    for s in range(0, n, 1 << 28):
        e = min(n, s + (1 << 28))
        d[s:e] = lut[(d[s:e] & 31).long()]
    d[n:] = 0
    ctx = M.DeviceContext(0)
    sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    t0 = time.time()
    ctx.make_sa(d, n, sa, two_stage=1)
    t1 = time.time()
    tm = ctx.timings()
    assert tm.bstar_suffixes > 0, "two-stage path declined"
    assert int(sa[0]) == n and ctx.validate_sa(d, n, sa) == 0
    print(f"\nn=2^31-2, 30 letters, two-stage: {t1 - t0:.3f}s (device {tm.total_ms:.1f} ms, {tm.bstar_suffixes} B* suffixes, induction {tm.other_ms:.1f} ms)")
    b1 = torch.empty(n, dtype=torch.uint8, device=dev)
    s1 = ctx.bwt_from_sa(d, n, sa, b1)
    del sa
    b2 = torch.empty(n, dtype=torch.uint8, device=dev)
    s2 = ctx.forward_bwt(d, n, b2, two_stage=1)
    assert ctx.timings().bstar_suffixes > 0
    assert s1 == s2 and torch.equal(b1, b2)
