#!/usr/bin/env python3
"""Two-stage vs sort-all on REAL files of this image (source code: .py/.h/.hpp/.txt/.rst/.json under site-packages and ROCm):
python tools/gpu_real_corpus.py [MiB]   (repetitive, with duplicated files: what the synthetic text generator is not)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import msufsort_amd as M  # noqa: E402

want = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
roots = ["/opt/rocm/include", "/usr/local/lib/python3.10/dist-packages", "/usr/lib/python3.10", "/usr/share/doc"]
exts = (".py", ".h", ".hpp", ".txt", ".rst", ".md", ".json", ".cuh", ".pyi", ".cpp", ".c", ".html")
buf = bytearray()
for r in roots:
    for dp, dn, fn in os.walk(r):
        dn.sort()
        for f in sorted(fn):
            if f.endswith(exts):
                try:
                    with open(os.path.join(dp, f), "rb") as fh:
                        b = fh.read()
                except OSError:
                    continue
                buf += b.replace(b"\x00", b" ")
                if len(buf) >= want:
                    break
        if len(buf) >= want:
            break
    if len(buf) >= want:
        break
t = np.frombuffer(bytes(buf[:want]), dtype=np.uint8)
n = t.size
vals = len(np.unique(t[: 1 << 24]))
print(f"corpus: {n >> 20} MiB, {vals} byte values in the first 16 MiB", flush=True)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
d[:n] = torch.from_numpy(t).cuda()
sa0 = torch.empty(n + 1, dtype=torch.int32, device="cuda")
sa1 = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, 0)
for mode, out in ((-1, sa0), (1, sa1), (0, sa1)):
    best = 1e9
    for r in range(2):
        ctx.make_sa(d, n, out, two_stage=mode, verbose=(1 if (mode == 1 and r == 0) else 0))
        best = min(best, ctx.timings().total_ms)
    tm = ctx.timings()
    print(f"two_stage={mode:2d}: {best:8.2f} ms  (taken: {tm.bstar_suffixes > 0}, rounds {tm.rounds}, doubling rounds {tm.doubling_rounds}, induction {tm.other_ms:.2f} ms)", flush=True)
print("equal", bool(torch.equal(sa0, sa1)), "checker errors", ctx.validate_sa(d, n, sa1))
