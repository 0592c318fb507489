"""verbose build of random bytes (round statistics, segments the bucket sort handed back); usage: gpu_verbose_random.py <log2 n> [minus]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msufsort_amd as M
from msufsort_amd import gen
n = (1 << int(sys.argv[1])) - (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = gen.random_bytes(n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0)
ctx.make_sa(d, n, sa, verbose=1)
print("errors", ctx.validate_sa(d, n, sa), "ms", ctx.timings().total_ms)
