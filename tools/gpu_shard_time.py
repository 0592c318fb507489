#!/usr/bin/env python3
"""Per-shard build time on ONE GPU (what each rank of an N-GPU run computes before the all-gatherv)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import msufsort_amd as M
from msufsort_amd import gen
n = (1 << 30) - 1
workload = sys.argv[1] if len(sys.argv) > 1 else "random"
t = gen.GENERATORS[workload](n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
full = torch.empty(n + 1, dtype=torch.int32, device="cuda")
grp = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
for G in (1, 2, 4, 8):
    bounds = ctx.shard_bounds(d, n, G)
    times = []
    for g in range(G):
        lo, hi = bounds[g], bounds[g + 1]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.make_sa_shard_groups(d, n, full[lo:hi], grp[lo:hi], hi - lo, g, G, text_rounds=(8 if workload == 'random' else 0))
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        times.append(best * 1e3)
    ok = ctx.validate_sa(d, n, full) == 0
    print(f"G={G}: per-shard ms {['%.2f' % x for x in times]} max {max(times):.2f} valid {ok}", flush=True)
    tm = ctx.timings()
    print(f"      last shard device phases: total {tm.total_ms:.2f} hist {tm.hist16_ms:.2f} scatter0 {tm.scatter0_ms:.2f} partition {tm.scatter1_ms:.2f} sorts {tm.bucket_sort_ms:.2f} refine {tm.refine_ms:.2f}", flush=True)
