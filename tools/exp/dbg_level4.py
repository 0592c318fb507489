import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import msufsort_amd as M, oracle
from msufsort_amd import gen
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 64
r = gen.random_bytes(reps, 3) % nv + 100
t = np.empty(reps * 5, dtype=np.uint8)
t[0::5] = 65; t[1::5] = 66; t[2::5] = 67; t[3::5] = 68; t[4::5] = r
n = t.size
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
ctx.make_sa(d, n, sa, verbose=1)
print("checker errors", ctx.validate_sa(d, n, sa))
if oracle.have_reference():
    print("reference identical", bool((sa.cpu().numpy() == oracle.ref_make_suffix_array(t, 8)).all()))
