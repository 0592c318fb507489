import sys, time, torch
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import msufsort_amd as M
from msufsort_amd import gen
n = (1 << 30) - 1
t = gen.text_bytes(n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
b = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx = M.DeviceContext(0, 0)
s = ctx.forward_bwt(d, n, b)
inv = torch.empty(n, dtype=torch.uint8, device="cuda")
for r in range(3):
    try:
        ctx.inverse_bwt(b, n, s, inv)
    except Exception as e:
        print("err", str(e)[:80])
    tm = ctx.timings(); print("inverse device ms", tm.ibwt_total_us / 1e3, "walk", tm.ibwt_walk_us / 1e3, flush=True)
