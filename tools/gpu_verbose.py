#!/usr/bin/env python3
"""Per-round statistics of one build (verbose=1): python tools/gpu_verbose.py <workload> <n> [text_rounds]"""
import sys

import torch

sys.path.insert(0, ".")
import msufsort_amd as M  # noqa: E402
from msufsort_amd import gen  # noqa: E402

w, n = sys.argv[1], int(sys.argv[2])
tr = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = gen.GENERATORS[w](n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
ctx.make_sa(d, n, sa)
ctx.make_sa(d, n, sa, verbose=1, text_rounds=tr)
tm = ctx.timings()
print("total_ms", tm.total_ms, "hist", tm.hist16_ms, "s0", tm.scatter0_ms, "s1", tm.scatter1_ms, "bucket", tm.bucket_sort_ms, "refine", tm.refine_ms)
