"""BASELINE config 5's own stream (gen.dna_tandem_bytes) at n = 2^LOG on one GPU through the wide engine, verbose: where the
distributed doubling spends its time.  usage: gpu_cfg5_verbose.py [log2 n = 33] [logical shards = 32]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msufsort_amd as M
from msufsort_amd import gen
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 33
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = 1 << lg
t = gen.dna_tandem_bytes(n, 9)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
for s in range(0, n, 1 << 30):
    d[s: s + (1 << 30)] = torch.from_numpy(t[s: s + (1 << 30)]).cuda()
del t
ctx = M.DeviceContext(0)
sa = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.make_sa_i64(d, n, sa, n_shards=shards, force_wide=True)
torch.cuda.synchronize(); t0 = time.time()
ctx.make_sa_i64(d, n, sa, n_shards=shards, force_wide=True, verbose=1)
torch.cuda.synchronize(); t1 = time.time()
tm = ctx.timings()
print(f"second build {t1 - t0:.2f} s, doubling {tm.other_ms:.0f} ms in {tm.doubling_rounds} steps, {tm.logical_shards} shards, progressions {tm.progression_suffixes}")
