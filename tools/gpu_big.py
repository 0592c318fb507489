#!/usr/bin/env python3
"""One-off: n = 2^31 - 2 (the int32 interface limit) uniform random bytes on one MI355X, checked on device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msufsort_amd as M
from msufsort_amd import gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 31) - 2
t0 = time.time(); t = gen.random_bytes(n, 777); print(f"generated n={n} in {time.time()-t0:.1f}s", flush=True)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
print(f"free/total HBM GiB: {[x/2**30 for x in torch.cuda.mem_get_info()]}", flush=True)
for rep in range(2):
    t0 = time.time(); ctx.make_sa(d, n, sa, verbose=1 if rep == 0 else 0); dt = time.time() - t0
print(f"SA: {dt*1e3:.1f} ms, {n/dt/1e6:.0f} MB/s", flush=True)
err = ctx.validate_sa(d, n, sa)
print("on-device checker errors:", err, "SA[0] =", int(sa[0]) & 0xffffffff, flush=True)
sys.exit(1 if err else 0)
