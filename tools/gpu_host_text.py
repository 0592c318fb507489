#!/usr/bin/env python3
"""Host-pointer legs on one workload, checked row by row against a device-resident build: msufsort_hip_make_sa_multi into a fresh
np.empty (with MSUFSORT_HIP_HOST_TRACE=1 the library's own timeline) and msufsort_hip_forward_bwt in place.
usage: gpu_host_text.py [workload] [n] [reps] [ops=sa,fbwt]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import msufsort_amd as M
from msufsort_amd import _lib, gen
from msufsort_amd.api import _opts
workload = sys.argv[1] if len(sys.argv) > 1 else "text"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 30) - 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ops = sys.argv[4].split(",") if len(sys.argv) > 4 else ["sa", "fbwt"]
t = gen.GENERATORS[workload](n, 3 if workload == "text" else (12345 if workload == "random" else 9))
L = _lib.lib()
dev = torch.device("cuda")
d = torch.zeros(n + 64, dtype=torch.uint8, device=dev); d[:n] = torch.from_numpy(t).to(dev)
ctx = M.DeviceContext(0)
ref = torch.empty(n + 1, dtype=torch.int32, device=dev)
ctx.make_sa(d, n, ref)
print(f"device build: {ctx.timings().total_ms:.2f} ms, checker errors {ctx.validate_sa(d, n, ref)}", flush=True)
if "sa" in ops:
    for r in range(reps + 1):
        sa = np.empty(n + 1, dtype=np.int32)
        src = t.copy()
        o = _opts(n_shards=0); dv = (C.c_int32 * 1)(0)
        t0 = time.perf_counter()
        _lib.check(L.msufsort_hip_make_sa_multi(dv, 1, src.ctypes.data, n, sa.ctypes.data, 4, C.byref(o), None), "make_sa_multi")
        dt = time.perf_counter() - t0
        same = bool(torch.equal(torch.from_numpy(sa).to(dev), ref))
        print(f"{workload} sa rep {r}: {dt * 1e3:.1f} ms; rows equal to the device build: {same}", flush=True)
        del sa
if "fbwt" in ops:
    bref = torch.empty(n, dtype=torch.uint8, device=dev)
    sref = ctx.bwt_from_sa(d, n, ref, bref)
    for r in range(reps + 1):
        buf = t.copy(); s = C.c_int64(0); o = _opts(n_shards=0)
        t0 = time.perf_counter()
        _lib.check(L.msufsort_hip_forward_bwt(buf.ctypes.data, n, C.byref(s), C.byref(o)), "forward_bwt")
        dt = time.perf_counter() - t0
        same = bool(torch.equal(torch.from_numpy(buf).to(dev), bref)) and s.value == sref
        print(f"{workload} fbwt rep {r}: {dt * 1e3:.1f} ms; bytes + sentinel equal to the device build: {same}", flush=True)
