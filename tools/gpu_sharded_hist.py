"""What the sharded histogram saves a rank of an N-GPU job (one GPU here: the two collectives are done by hand and not timed).
python tools/gpu_sharded_hist.py [n] [parts]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import msufsort_amd as M
from msufsort_amd import gen

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30) - 1
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda")
t = gen.random_bytes(n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
d[:n] = torch.from_numpy(t).to(dev)
ctx = M.DeviceContext(0)
bounds = ctx.shard_bounds(d, n, parts)
rows_max = max(b - a for a, b in zip(bounds, bounds[1:]))
sl = torch.empty(rows_max, dtype=torch.int32, device=dev)
h = torch.empty(65536, dtype=torch.int64, device=dev)
# totals and stripe sums of all parts, once (what the collectives would deliver)
hs, blocks, geo = [], [], []
for p in range(parts):
    geo.append(ctx.hist_part(d, n, p, parts, h))
    hs.append(h.clone())
hsum = torch.stack(hs).sum(0)
per = -(-geo[0][0] // parts)
for p in range(parts):
    ctx.hist_part(d, n, p, parts, h)
    s = torch.empty((parts, per, 256), dtype=torch.int32, device=dev)
    assert ctx.hist_plan(d, n, parts, hsum, s) == bounds
    blocks.append(s)
for g in (0, parts // 2):
    mine = torch.cat([blocks[p][g, :geo[p][2] - geo[p][1]] for p in range(parts)]).contiguous()
    rep, shd = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        a = time.perf_counter()
        ctx.make_sa_shard(d, n, sl, rows_max, g, parts)
        torch.cuda.synchronize()
        rep.append((time.perf_counter() - a) * 1e3)
        ref = sl[:bounds[g + 1] - bounds[g]].clone()
        torch.cuda.synchronize()
        a = time.perf_counter()
        ctx.hist_part(d, n, g, parts, h)
        s = torch.empty((parts, per, 256), dtype=torch.int32, device=dev)
        ctx.hist_plan(d, n, parts, hsum, s)
        ctx.hist_install(g, mine)
        b = time.perf_counter()
        ctx.make_sa_shard(d, n, sl, rows_max, g, parts)
        torch.cuda.synchronize()
        shd.append(((time.perf_counter() - a) * 1e3, (b - a) * 1e3))
        assert torch.equal(ref, sl[:bounds[g + 1] - bounds[g]])
    print(f"n {n} shard {g} of {parts}: replicated histogram {min(rep):.3f} ms per build; sharded {min(x[0] for x in shd):.3f} ms "
          f"(hist_part + plan + install {min(x[1] for x in shd):.3f} ms; collectives not included)", flush=True)
