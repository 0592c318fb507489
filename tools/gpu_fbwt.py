import sys, time, torch
sys.path.insert(0, ".")
import msufsort_amd as M
from msufsort_amd import gen
n = (1 << 30) - 1
t = gen.text_bytes(n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
b = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx = M.DeviceContext(0, 0)
for r in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    s = ctx.forward_bwt(d, n, b)
    torch.cuda.synchronize(); print("forward_bwt wall ms", (time.time() - t0) * 1e3, "sa device ms", ctx.timings().total_ms, flush=True)
inv = torch.empty(n, dtype=torch.uint8, device="cuda")
for r in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    ctx.inverse_bwt(b, n, s, inv)
    torch.cuda.synchronize(); print("inverse wall ms", (time.time() - t0) * 1e3, flush=True)
