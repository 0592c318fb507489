#!/usr/bin/env python3
"""Duplicate-heavy random inputs at sizes where the 17-bit levels feed the 18,432-record shape of k_sort_bits (two-byte buckets
above 18 K): every segment's dirty list overflows - the path whose flag race round 4 fixed - checked on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msufsort_amd as M
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_big import _random_gpu

dev = torch.device("cuda")
ctx = M.DeviceContext(0)
for copies, part_mib in ((2, 640), (3, 430), (2, 990)):
    part = part_mib << 20
    base = _random_gpu(part, 500 + copies, dev)
    n = part * copies
    d = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    for k in range(copies):
        d[k * part:(k + 1) * part] = base[:part]
    del base
    sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    t0 = time.time()
    ctx.make_sa(d, n, sa)
    torch.cuda.synchronize()
    dt = time.time() - t0
    tm = ctx.timings()
    print(f"{copies} x {part_mib} MiB random: {dt:.2f} s, radix bits {tm.radix_bits}, rounds {tm.rounds} (doubling {tm.doubling_rounds}), checker errors {ctx.validate_sa(d, n, sa)}", flush=True)
    del d, sa
    ctx.trim(); torch.cuda.empty_cache()
