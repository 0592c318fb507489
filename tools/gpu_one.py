"""One workload, a few builds - the process rocprofv3 wraps.  python tools/gpu_one.py workload n two_stage reps [ops]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import msufsort_amd as M
from msufsort_amd import gen

workload, n, two_stage, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ops = sys.argv[5].split(",") if len(sys.argv) > 5 else ["sa"]
dev = torch.device("cuda")
t = gen.GENERATORS[workload](n, 3 if workload == "text" else (12345 if workload == "random" else 9))
d = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
d[:n] = torch.from_numpy(t).to(dev)
ctx = M.DeviceContext(0)
sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
bwt = torch.empty(n, dtype=torch.uint8, device=dev)
inv = torch.empty(n, dtype=torch.uint8, device=dev)
lcp = torch.empty(n, dtype=torch.int32, device=dev) if "lcp" in ops else None
for r in range(reps):
    ctx.make_sa(d, n, sa, two_stage=two_stage)
    tm = ctx.timings()
    if "ibwt" in ops:
        s = ctx.bwt_from_sa(d, n, sa, bwt)
        ctx.inverse_bwt(bwt, n, s, inv)
    if "lcp" in ops:
        ctx.lcp(d, n, sa, lcp)
    print(f"build {r}: {tm.total_ms:.2f} ms", flush=True)
print("errors", ctx.validate_sa(d, n, sa))
