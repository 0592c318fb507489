#!/usr/bin/env python3
"""Pathological inputs at scale: one symbol, period 2, period 3 with a defect - checked on device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import msufsort_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
cases = {"allA": np.full(n, 65, np.uint8), "ab": np.resize(np.frombuffer(b"ab", np.uint8), n),
         "abc+defect": np.resize(np.frombuffer(b"abc", np.uint8), n).copy(), "zeros": np.zeros(n, np.uint8)}
cases["abc+defect"][n // 2] = 120
ctx = M.DeviceContext(0, n)
for name, t in cases.items():
    d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    t0 = time.time(); ctx.make_sa(d, n, sa); dt = time.time() - t0
    tm = ctx.timings()
    t0 = time.time(); err = ctx.validate_sa(d, n, sa); dv = time.time() - t0
    print(f"{name}: n={n} SA {dt*1e3:.1f} ms ({n/dt/1e6:.0f} MB/s) rounds {tm.rounds} doubling {tm.doubling_rounds} checker errors {err} ({dv:.1f}s)", flush=True)
