#!/usr/bin/env python3
"""Two-stage build (B* sort + induction) against the sort-all path: python tools/gpu_two_stage.py <workload> <n> [reps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import msufsort_amd as M  # noqa: E402
from msufsort_amd import gen  # noqa: E402

w, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t = gen.GENERATORS[w](n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
d[:n] = torch.from_numpy(t).cuda()
sa0 = torch.empty(n + 1, dtype=torch.int32, device="cuda")
sa1 = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, 0)
for r in range(reps):
    ctx.make_sa(d, n, sa0, two_stage=-1)
print("sort-all ms", ctx.timings().total_ms, flush=True)
for r in range(reps):
    t0 = time.time()
    ctx.make_sa(d, n, sa1, two_stage=1, verbose=1 if r == reps - 1 else 0)
    torch.cuda.synchronize()
    print("two-stage wall ms", (time.time() - t0) * 1e3, "device ms", ctx.timings().total_ms, flush=True)
eq = bool(torch.equal(sa0, sa1))
print("equal", eq)
if not eq:
    bad = (sa0 != sa1).nonzero().flatten()
    print("mismatches", bad.numel(), "first rows", bad[:8].tolist(), sa0[bad[:8]].tolist(), sa1[bad[:8]].tolist())
    sys.exit(1)
