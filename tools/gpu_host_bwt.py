"""Host-pointer forward BWT of 2^30 - 1 random bytes, streamed (msufsort_hip_forward_bwt_multi) against the one-device call; with
MSUFSORT_HIP_HOST_TRACE=1 the library's own timeline.  python tools/gpu_host_bwt.py [n] [reps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from msufsort_amd import _lib, gen
from msufsort_amd.api import _opts
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30) - 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = gen.random_bytes(n, 12345)
L = _lib.lib()
for name in ("multi", "single"):
    ms = []
    for r in range(reps + 1):
        buf = t.copy(); s = C.c_int64(0); o = _opts(n_shards=0); dv = (C.c_int32 * 1)(0)
        t0 = time.perf_counter()
        if name == "multi": _lib.check(L.msufsort_hip_forward_bwt_multi(dv, 1, buf.ctypes.data, n, C.byref(s), C.byref(o), None), "fbwt multi")
        else: _lib.check(L.msufsort_hip_forward_bwt(buf.ctypes.data, n, C.byref(s), C.byref(o)), "fbwt")
        if r: ms.append(round((time.perf_counter() - t0) * 1e3, 2))
        if name == "multi": first = buf
        else: same = bool((buf == first).all())
    print(name, ms, "sentinel", s.value, flush=True)
print("equal", same)
