#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (never bench.py's `value`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import msufsort_amd as M
from msufsort_amd import gen
for n in (1 << 28, (1 << 30) - 1):
    t = gen.random_bytes(n, 12345)
    eng = M.DeviceContext(0, n)      # keeps the workspace so the timing below is transfers + kernels
    import ctypes as C
    from msufsort_amd import _lib
    L = _lib.lib(); sa = np.empty(n + 1, np.int32); o = _lib.Opts()
    for pinmode in ("pin", "nopin"):
        if pinmode == "nopin": os.environ["MSUFSORT_HIP_NO_PIN"] = "1"
        else: os.environ.pop("MSUFSORT_HIP_NO_PIN", None)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            _lib.check(L.msufsort_hip_make_sa_i32_ctx(eng._h, t.ctypes.data, n, sa.ctypes.data, C.byref(o)), "sa")
            best = min(best, time.perf_counter() - t0)
        print(f"n={n} {pinmode}: host-pointer make_sa {best*1e3:.1f} ms = {n/best/1e6:.0f} MB/s (H2D n + D2H 4n bytes + build)", flush=True)
    assert sa[0] == n
