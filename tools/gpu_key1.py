"""A/B of "key of the first gather round from the sequential pass" (DESIGN 1.4): MSUFSORT_HIP_KEY1=-1 (every gather round gathers,
the round-4 behaviour) against the default, on the text / DNA workloads, with the per-round lines of one verbose build each.
Run on the GPU box: python tools/gpu_key1.py [size]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import msufsort_amd as M
from msufsort_amd import gen

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30) - 1
dev = torch.device("cuda")
for workload, two_stage in (("text", 0), ("text", -1), ("dna", 0), ("dna", -1), ("dna_tandem", 0)):
    nn = n if workload != "dna_tandem" else min(n, 1 << 28)
    t = gen.GENERATORS[workload](nn, 3 if workload == "text" else 9)
    d = torch.zeros(nn + 64, dtype=torch.uint8, device=dev)
    d[:nn] = torch.from_numpy(t).to(dev)
    del t
    ctx = M.DeviceContext(0)
    out = {}
    for key1 in ("-1", "0"):
        os.environ["MSUFSORT_HIP_KEY1"] = key1
        sa = torch.empty(nn + 1, dtype=torch.int32, device=dev)
        best = None
        for rep in range(4):
            ctx.make_sa(d, nn, sa, two_stage=two_stage, verbose=1 if rep == 3 else 0)
            tm = ctx.timings()
            if rep and (best is None or tm.total_ms < best[0]):
                best = (tm.total_ms, tm.hist16_ms, tm.scatter0_ms, tm.scatter1_ms, tm.bucket_sort_ms, tm.refine_ms, tm.other_ms, tm.gathered_records, tm.rounds)
        errs = ctx.validate_sa(d, nn, sa)
        out[key1] = sa
        print(f"RESULT {workload} n={nn} two_stage={two_stage} KEY1={key1}: total {best[0]:.2f} ms (hist {best[1]:.2f} scatter0 {best[2]:.2f} level1 {best[3]:.2f} "
              f"round-0 sorts {best[4]:.2f} later rounds {best[5]:.2f} induction {best[6]:.2f}), gathered records {best[7]}, rounds {best[8]}, checker errors {errs}", flush=True)
    print(f"RESULT {workload} two_stage={two_stage}: rows equal {bool(torch.equal(out['-1'], out['0']))}", flush=True)
    del out, sa, d, ctx
    torch.cuda.empty_cache()
