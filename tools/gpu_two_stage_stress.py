#!/usr/bin/env python3
"""Repeats the two-stage build and compares every result with the sort-all rows (timing-dependent faults of the single-pass
induction levels would show as a difference or a reported failure): python tools/gpu_two_stage_stress.py <workload> <MiB> <reps>"""
import sys

import torch

sys.path.insert(0, ".")
import msufsort_amd as M  # noqa: E402
from msufsort_amd import gen  # noqa: E402

w, mib, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = (mib << 20) - 1
t = gen.GENERATORS[w](n, 4242)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
d[:n] = torch.from_numpy(t).cuda()
ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, 0)
ctx.make_sa(d, n, ref, two_stage=-1)
bad = 0
for r in range(reps):
    sa.zero_()
    ctx.make_sa(d, n, sa, two_stage=1)
    assert ctx.timings().bstar_suffixes > 0
    if not torch.equal(sa, ref):
        bad += 1
        print("MISMATCH in repetition", r, flush=True)
print(f"{w} {mib} MiB: {reps} two-stage builds, {bad} differ from the sort-all rows")
sys.exit(1 if bad else 0)
