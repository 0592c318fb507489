#!/usr/bin/env python3
"""Adversarial inputs for the suffix-array build (on-device checker = exact; small sizes also against the reference):
repeats at every scale, Fibonacci / Thue-Morse / period-doubling words, skewed alphabets, zero and 0xFF runs."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import msufsort_amd as M  # noqa: E402
import oracle  # noqa: E402
from msufsort_amd import gen  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
SMALL = 1 << 20


def fib(n):
    a, b = np.array([98], np.uint8), np.array([97], np.uint8)
    while b.size < n:
        a, b = b, np.concatenate([b, a])
    return b[:n].copy()


def thue(n):
    i = np.arange(n, dtype=np.uint64)
    x = i.copy()
    for s in (32, 16, 8, 4, 2, 1):
        x ^= x >> np.uint64(s)
    return (97 + (x & np.uint64(1))).astype(np.uint8)


def period_doubling(n):
    i = np.arange(1, n + 1, dtype=np.uint64)
    tz = np.zeros(n, dtype=np.uint64)
    x = i.copy()
    for _ in range(40):
        even = (x & np.uint64(1)) == 0
        tz += even
        x = np.where(even, x >> np.uint64(1), x)
    return (97 + (tz & np.uint64(1))).astype(np.uint8)


def cases(n):
    r = gen.random_bytes(n, 5)
    yield "random x3 copies", np.concatenate([r[: n // 3]] * 3)
    yield "random x8 copies", np.concatenate([r[: n // 8]] * 8)
    t = gen.text_bytes(n // 2, 9)
    yield "text + copy", np.concatenate([t, t])
    yield "text + reversed copy", np.concatenate([t, t[::-1]])
    yield "fibonacci", fib(n)
    yield "thue-morse", thue(n)
    yield "period-doubling", period_doubling(n)
    s = np.where(r < 3, r, 97).astype(np.uint8)
    yield "99% 'a'", s
    z = r.copy(); z[n // 4: n // 2] = 0; z[-(n // 16):] = 0
    yield "zero runs (middle + tail)", z
    f = r.copy(); f[n // 4: n // 2] = 255; f[-(n // 16):] = 255
    yield "0xFF runs (middle + tail)", f
    yield "two symbols random", (97 + (r & 1)).astype(np.uint8)
    blk = gen.random_bytes(4099, 6)
    yield "4099-byte block tiled", np.tile(blk, n // 4099 + 1)[:n].copy()
    d = gen.dna_tandem_bytes(n, 3) if hasattr(gen, "dna_tandem_bytes") else gen.dna_bytes(n, 3)
    yield "dna tandem + copy of its first half", np.concatenate([d, d[: n // 2]])


bad = 0
ctx = M.DeviceContext(0, 2 * N)
for size, ref in ((SMALL, True), (N, False)):
    for name, t in cases(size):
        n = t.size
        d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        d[:n] = torch.from_numpy(t).cuda()
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        t0 = time.time()
        try:
            ctx.make_sa(d, n, sa)
            torch.cuda.synchronize()
            dt = time.time() - t0
            err = ctx.validate_sa(d, n, sa)
            tm = ctx.timings()
            msg = f"{dt*1e3:8.1f} ms rounds {tm.rounds:3d} (doubling {tm.doubling_rounds:2d}) checker errors {err}"
            if ref and oracle.have_reference():
                want = oracle.ref_make_suffix_array(t, 8)
                same = bool((sa.cpu().numpy() == want).all())
                msg += f" reference identical: {same}"
                err += 0 if same else 1
        except Exception as e:  # noqa: BLE001
            msg, err = f"FAILED: {e}", 1
        bad += 1 if err else 0
        print(f"n={n:>10d} {name:38s} {msg}", flush=True)
print("RESULT", "PASS" if bad == 0 else f"FAIL ({bad})")
sys.exit(0 if bad == 0 else 1)
