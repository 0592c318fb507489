#!/usr/bin/env python3
"""Latency of device-resident builds for small and medium inputs (one context, warm)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import msufsort_amd as M
from msufsort_amd import gen
ctx = M.DeviceContext(0, 1 << 26)
for kind in ("random", "text"):
    for n in (1 << 12, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
        t = gen.GENERATORS[kind](n, 3)
        d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        ctx.make_sa(d, n, sa); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); ctx.make_sa(d, n, sa); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        tm = ctx.timings()
        print(f"{kind:7s} n={n:>9d}: {best*1e3:8.3f} ms wall ({n/best/1e6:9.1f} MB/s)  device {tm.total_ms:7.3f} ms  rounds {tm.rounds}", flush=True)
