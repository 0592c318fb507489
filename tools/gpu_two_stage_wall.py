#!/usr/bin/env python3
"""WALL time of make_sa with the two-stage path off / default / forced (a declined attempt is not in the device timings of
the build that follows it): python tools/gpu_two_stage_wall.py <workload|corpus> <MiB>"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import msufsort_amd as M  # noqa: E402
from msufsort_amd import gen  # noqa: E402

w, mib = sys.argv[1], int(sys.argv[2])
n = mib << 20
if w == "corpus":
    roots = ["/opt/rocm/include", "/usr/local/lib/python3.10/dist-packages", "/usr/lib/python3.10", "/usr/share/doc"]
    exts = (".py", ".h", ".hpp", ".txt", ".rst", ".md", ".json", ".cuh", ".pyi", ".cpp", ".c", ".html")
    buf = bytearray()
    for r in roots:
        for dp, dn, fn in os.walk(r):
            dn.sort()
            for f in sorted(fn):
                if f.endswith(exts) and len(buf) < n:
                    try:
                        buf += open(os.path.join(dp, f), "rb").read().replace(b"\x00", b" ")
                    except OSError:
                        pass
    t = np.frombuffer(bytes(buf[:n]), dtype=np.uint8).copy()
    n = t.size
else:
    t = gen.GENERATORS[w](n, 12345)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, 0)
for mode in (-1, 0, 1):
    best = 1e9
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.make_sa(d, n, sa, two_stage=mode)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"{w} {n >> 20} MiB two_stage={mode:2d}: wall {best:8.2f} ms (taken: {ctx.timings().bstar_suffixes > 0})", flush=True)
