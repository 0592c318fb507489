#!/usr/bin/env python3
"""Race hunt: the same input built again and again must give the same rows, bit for bit (the rows are unique, so any difference
is a race or an uninitialised read).  Random bytes at the headline size and at two 17-bit sizes, text through the two-stage
build (tile hand-out by handshake), DNA with tandem repeats (progressions + doubling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msufsort_amd as M
from msufsort_amd import gen
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_big import _random_gpu

dev = torch.device("cuda")
ctx = M.DeviceContext(0)


def run(name, d, n, reps, **kw):
    ref = torch.empty(n + 1, dtype=torch.int32, device=dev)
    sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ctx.make_sa(d, n, ref, **kw)
    err = ctx.validate_sa(d, n, ref)
    diff = 0
    for _ in range(reps):
        sa.fill_(-1)
        ctx.make_sa(d, n, sa, **kw)
        diff += 0 if torch.equal(sa, ref) else 1
    print(f"{name}: n={n}, checker errors {err}, {reps} more builds, {diff} differ", flush=True)
    return err + diff


bad = 0
for mib, reps in ((1024, 20), (300, 10), (1300, 6)):
    n = (mib << 20) - 1
    d = _random_gpu(n + 64, 4000 + mib, dev); d[n:] = 0
    bad += run(f"random {mib} MiB", d, n, reps)
    del d; torch.cuda.empty_cache()
n = (1 << 30) - 1
t = gen.text_bytes(n, 3)
d = torch.zeros(n + 64, dtype=torch.uint8, device=dev); d[:n] = torch.from_numpy(t).to(dev)
bad += run("text 1 GiB (two-stage)", d, n, 8)
del d, t; torch.cuda.empty_cache()
n = 1 << 28
t = gen.dna_tandem_bytes(n, 9)
d = torch.zeros(n + 64, dtype=torch.uint8, device=dev); d[:n] = torch.from_numpy(t).to(dev)
bad += run("tandem DNA 256 MiB", d, n, 6)
print("RESULT", "PASS" if bad == 0 else f"FAIL ({bad})")
sys.exit(0 if bad == 0 else 1)
