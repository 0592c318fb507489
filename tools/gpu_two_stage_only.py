#!/usr/bin/env python3
"""Only the two-stage build, for profiling: python tools/gpu_two_stage_only.py <workload> <n> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import msufsort_amd as M  # noqa: E402
from msufsort_amd import gen  # noqa: E402

w, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t = gen.GENERATORS[w](n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
d[:n] = torch.from_numpy(t).cuda()
sa1 = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, 0)
for r in range(reps):
    ctx.make_sa(d, n, sa1, two_stage=1, verbose=1 if r == reps - 1 else 0)
    print("device ms", ctx.timings().total_ms, "induction ms", ctx.timings().other_ms, flush=True)
