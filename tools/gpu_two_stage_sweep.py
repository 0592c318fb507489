#!/usr/bin/env python3
"""Crossover of the two-stage build against the sort-all path: python tools/gpu_two_stage_sweep.py <workload> <MiB> [<MiB> ...]"""
import sys

import torch

sys.path.insert(0, ".")
import msufsort_amd as M  # noqa: E402
from msufsort_amd import gen  # noqa: E402

w = sys.argv[1]
ctx = M.DeviceContext(0, 0)
for mib in [int(x) for x in sys.argv[2:]]:
    n = mib << 20
    t = gen.GENERATORS[w](n, 12345)
    d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    d[:n] = torch.from_numpy(t).cuda()
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    res = {}
    for mode in (-1, 1):
        best = 1e9
        for r in range(3):
            ctx.make_sa(d, n, sa, two_stage=mode)
            best = min(best, ctx.timings().total_ms)
        res[mode] = best
    tm = ctx.timings()
    print(f"{w} {mib} MiB: sort-all {res[-1]:.2f} ms, two-stage {res[1]:.2f} ms (induction {tm.other_ms:.2f}, launches {tm.induction_launches})", flush=True)
    del d, sa
