"""k_sort_bits at the class boundaries: random bytes whose two-byte buckets sit around the limits of its two shapes (4352 / 17408
records) and of the size classes (4608 / 18432), every build checked on the device (the suffix array is unique)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msufsort_amd as M
from msufsort_amd import gen
ctx = M.DeviceContext(0)
for mib, seed in ((270, 3), (285, 4), (296, 5), (302, 6), (1040, 7), (1075, 8), (1090, 9), (1120, 10), (1180, 11), (1300, 12)):
    n = (mib << 20) + seed * 7919
    t = gen.random_bytes(n, seed)
    d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, sa)
    ctx.make_sa(d, n, sa, verbose=1)
    tm = ctx.timings()
    print(f"{mib} MiB: mean bucket {n / 65536:.0f}, {tm.total_ms:.2f} ms (bucket sort {tm.bucket_sort_ms:.2f}), errors {ctx.validate_sa(d, n, sa)}", flush=True)
    del d, sa
