#!/usr/bin/env python3
"""Host-pointer entry point into a FRESH result array (np.empty: untouched pages), with the library's own timeline
(MSUFSORT_HIP_HOST_TRACE=1).  usage: gpu_host_fresh.py [n] [workload] [reps]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, ".")
from msufsort_amd import _lib, gen
from msufsort_amd.api import _opts
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30) - 1
workload = sys.argv[2] if len(sys.argv) > 2 else "random"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t = gen.GENERATORS[workload](n, 12345)
L = _lib.lib()
for r in range(reps):
    sa = np.empty(n + 1, dtype=np.int32)
    o = _opts(n_shards=int(os.environ.get("SHARDS", "0"))); dv = (C.c_int32 * 1)(0)
    t0 = time.perf_counter()
    _lib.check(L.msufsort_hip_make_sa_multi(dv, 1, t.ctypes.data, n, sa.ctypes.data, 4, C.byref(o), None), "make_sa_multi")
    dt = time.perf_counter() - t0
    print(f"{workload} rep {r}: {dt * 1e3:.1f} ms; SA[0]={sa[0]} SA[1]={sa[1]} SA[n]={sa[n]}", flush=True)
    del sa
