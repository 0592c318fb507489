import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import msufsort_amd as M
from msufsort_amd import gen
kind = sys.argv[1]; n = int(sys.argv[2])
t = gen.GENERATORS[kind](n, 3)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
ctx.make_sa(d, n, sa, verbose=1)
