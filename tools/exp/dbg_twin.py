import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import msufsort_amd as M
from msufsort_amd import gen
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 25
base = gen.random_bytes(nu, 77)
t = np.concatenate([base, base]); n = t.size
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
try:
    ctx.make_sa(d, n, sa, verbose=1)
    print("errors", ctx.validate_sa(d, n, sa))
except Exception as e:
    print("FAILED", e)
