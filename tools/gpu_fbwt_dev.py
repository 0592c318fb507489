#!/usr/bin/env python3
"""Device-resident forward transform as one call (msufsort_hip_forward_bwt_dev), wall clock around the call.  gpu_fbwt_dev.py workload n reps"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import msufsort_amd as M
from msufsort_amd import gen
workload, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
t = gen.GENERATORS[workload](n, 3 if workload == "text" else (12345 if workload == "random" else 9))
dev = torch.device("cuda")
d = torch.zeros(n + 64, dtype=torch.uint8, device=dev); d[:n] = torch.from_numpy(t).to(dev)
ctx = M.DeviceContext(0)
b = torch.empty(n, dtype=torch.uint8, device=dev)
ms = []
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); s = ctx.forward_bwt(d, n, b); ms.append(round((time.perf_counter() - t0) * 1e3, 2))
sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
ctx.make_sa(d, n, sa, two_stage=-1)
b0 = torch.empty(n, dtype=torch.uint8, device=dev)
s0 = ctx.bwt_from_sa(d, n, sa, b0)
print(workload, "forward_bwt_dev ms", ms, "sentinel", s, "equal to the transform of a sort-all build:", bool(s == s0 and torch.equal(b, b0)), flush=True)
