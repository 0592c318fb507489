"""Round 4: device time of the suffix-array build of uniform random bytes from 256 MiB to 2^31 - 2 (per byte: is it flat?),
with the level structure that ran (16 or 17 radix bits, segments the bucket sort handed back), every build checked on the
device.  Inputs are generated on the GPU (splitmix64, the same stream as gen.random_bytes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msufsort_amd as M
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_big import _random_gpu

sizes = [int(x) for x in sys.argv[1:]] or [256, 270, 285, 296, 320, 400, 512, 640, 768, 1024, 1120, 1180, 1300, 1536, 1800, 2047]
dev = torch.device("cuda")
ctx = M.DeviceContext(0)
for mib in sizes:
    n = min((mib << 20) + 7919, (1 << 31) - 2)
    d = _random_gpu(n + 64, 1000 + mib, dev)
    d[n:] = 0
    sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ctx.make_sa(d, n, sa)
    best = None
    for _ in range(3):
        ctx.make_sa(d, n, sa)
        tm = ctx.timings()
        if best is None or tm.total_ms < best.total_ms:
            best = tm
    tm = best
    err = ctx.validate_sa(d, n, sa)
    print(f"{mib} MiB: mean bucket {n / 65536:.0f}, {tm.total_ms:.2f} ms = {tm.total_ms / (n / 2**30):.2f} ms/GiB (hist {tm.hist16_ms:.2f} [17-bit {tm.hist17_ms:.2f}], scatter0 {tm.scatter0_ms:.2f}, "
          f"level 1 {tm.scatter1_ms:.2f}, bucket sort {tm.bucket_sort_ms:.2f}, refine {tm.refine_ms:.2f}), radix bits {tm.radix_bits}, handed back {tm.bucket_sort_handed_back}, errors {err}", flush=True)
    del d, sa
    torch.cuda.empty_cache()
