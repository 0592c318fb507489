"""verbose build (round statistics, segments the bucket sort handed back); usage: gpu_verbose_any.py <generator> <n> [two_stage]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msufsort_amd as M
from msufsort_amd import gen
w, n = sys.argv[1], int(sys.argv[2])
t = gen.GENERATORS[w](n, {"random": 12345, "text": 3, "dna": 2024, "dna_tandem": 9}[w])
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0)
ctx.make_sa(d, n, sa, two_stage=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
ctx.make_sa(d, n, sa, verbose=1, two_stage=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
tm = ctx.timings()
print("errors", ctx.validate_sa(d, n, sa), "ms", tm.total_ms, "rounds", tm.rounds, "doubling", tm.doubling_rounds, "fallbacks", tm.fallbacks)
