"""BASELINE config 5's workload on the one GPU - the process rocprofv3 wraps (kernel stats, PMC passes).
python tools/gpu_cfg5.py [log2 n = 33] [builds = 2] [shards = 32]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import msufsort_amd as M
from msufsort_amd import gen

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 33
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 32
n = 1 << lg
dev = torch.device("cuda")
t0 = time.time()
t = gen.dna_tandem_bytes(n, 9)
d = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
for s in range(0, n, 1 << 30):
    d[s:s + (1 << 30)] = torch.from_numpy(t[s:s + (1 << 30)]).to(dev)
del t
torch.cuda.synchronize()
print(f"generated + uploaded in {time.time() - t0:.1f} s", flush=True)
ctx = M.DeviceContext(0)
sa = torch.empty(n + 1, dtype=torch.int64, device=dev)
for r in range(builds):
    t1 = time.time()
    ctx.make_sa_i64(d, n, sa, n_shards=shards, verbose=int(os.environ.get("CFG5_VERBOSE", "0")))
    tm = ctx.timings()
    print(f"build {r}: wall {(time.time() - t1) * 1e3:.1f} ms, device {tm.total_ms:.1f} ms: hist {tm.hist16_ms:.1f} scatter0 {tm.scatter0_ms:.1f} level1 {tm.scatter1_ms:.1f} round-0 sorts {tm.bucket_sort_ms:.1f} "
          f"key rounds {tm.refine_ms:.1f} doubling {tm.other_ms:.1f} ({tm.doubling_rounds} steps, {tm.doubling_records} rows scanned), {tm.logical_shards} shards, "
          f"gathered {tm.gathered_records}", flush=True)
if os.environ.get("CFG5_CHECK", "1") != "0":
    ctx.trim()
    print("errors", ctx.validate_sa(d, n, sa, index_bytes=8))
