#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points on 2^30 - 1 random bytes (never `value` in bench.py):
msufsort_hip_make_sa_i32 (H2D, build, D2H one after the other) against msufsort_hip_make_sa_multi on one GPU
(key-range shards; every finished slice leaves for the host while the next one is sorted).  The output array is
allocated and touched once, outside the timed calls (a fresh 4 GiB numpy array costs ~0.3 s of page faults - the cost the
reference pays at msufsort.cpp:1754-1758)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from msufsort_amd import _lib, gen  # noqa: E402
from msufsort_amd.api import _opts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30) - 1
workload = sys.argv[2] if len(sys.argv) > 2 else "random"
t = gen.GENERATORS[workload](n, 12345)
print("workload:", workload)
sa = np.zeros(n + 1, dtype=np.int32)
L = _lib.lib()


def seq():
    o = _opts()
    _lib.check(L.msufsort_hip_make_sa_i32(t.ctypes.data, n, sa.ctypes.data, C.byref(o)), "make_sa_i32")


def multi(shards):
    def f():
        o = _opts(n_shards=shards)
        dv = (C.c_int32 * 1)(0)
        _lib.check(L.msufsort_hip_make_sa_multi(dv, 1, t.ctypes.data, n, sa.ctypes.data, 4, C.byref(o), None), "make_sa_multi")
    return f


for name, f in (("make_sa_i32 (H2D, build, D2H in sequence)", seq), ("make_sa_multi [0], 4 shards", multi(4)),
                ("make_sa_multi [0], 8 shards (default)", multi(0)), ("make_sa_multi [0], 16 shards", multi(16)),
                ("make_sa_multi [0], 32 shards", multi(32))):
    best = None
    for _ in range(4):
        sa[:8] = 0
        t0 = time.perf_counter()
        f()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{name}: {best * 1e3:.1f} ms = {n / best / 1e9:.2f} GB/s of input; SA[0]={sa[0]} SA[1]={sa[1]} SA[n]={sa[n]}")
