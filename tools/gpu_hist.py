#!/usr/bin/env python3
"""k_hist16 (+ k_reduce16) alone on 2^30 - 1 bytes of several shapes: random (optimistic pass only), all-'A' / period-2 /
text / DNA (recount pass with run and wave aggregation); checked against numpy on a sample."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import msufsort_amd as M  # noqa: E402
from msufsort_amd import gen  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30) - 1
ctx = M.DeviceContext(0)
h = torch.zeros(65536, dtype=torch.int32, device="cuda")
for kind in ("random", "all_a", "ab", "text", "dna"):
    if kind == "random":
        t = gen.random_bytes(n, 1)
    elif kind == "all_a":
        t = np.full(n, 65, np.uint8)
    elif kind == "ab":
        t = np.frombuffer((b"ab" * (n // 2 + 1))[:n], dtype=np.uint8)
    elif kind == "text":
        t = gen.text_bytes(min(n, 1 << 28), 3); t = np.tile(t, n // t.size + 1)[:n]
    else:
        t = gen.dna_bytes(n, 4)
    d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    d[:n] = torch.from_numpy(np.ascontiguousarray(t)).cuda()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.debug_hist16(d, n, h)
        best = min(best, time.perf_counter() - t0)
    tp = np.concatenate([t, np.zeros(1, np.uint8)]).astype(np.uint32)
    want = np.bincount((tp[:-1] << 8) | tp[1:], minlength=65536)
    ok = bool((h.cpu().numpy().astype(np.int64) == want).all())
    print(f"{kind}: hist16 of {n} bytes {best * 1e3:.3f} ms (host wall incl. 256 KiB copy) = {n / best / 1e9:.0f} GB/s, exact={ok}")
    del d
