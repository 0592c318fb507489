import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import msufsort_amd as M
from msufsort_amd import gen
n = int(sys.argv[1]); nsym = int(sys.argv[2]) if len(sys.argv) > 2 else 64
r = gen.random_bytes(n, 3)
t = (48 + (r % nsym)).astype(np.uint8)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
ctx.make_sa(d, n, sa); torch.cuda.synchronize()
t0 = time.time(); ctx.make_sa(d, n, sa); torch.cuda.synchronize(); dt = time.time() - t0
tm = ctx.timings()
print(f"{nsym} symbols n={n}: {dt*1e3:.1f} ms ({n/dt/1e6:.0f} MB/s) rounds {tm.rounds} errors {ctx.validate_sa(d, n, sa)}")
