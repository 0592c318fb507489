#!/usr/bin/env python3
"""BASELINE.json configs 3 and 4 at full size on one MI355X: 1 GiB (2^30-1) text SA + forward BWT,
then BWT -> inverse BWT round trip, all in HBM, checked with size-independent properties
(on-device order+permutation checker, sentinel consistency, bit-exact round trip) and - for the
SA - against the unmodified reference when it is present (oracle/_ref, 32 threads)."""
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

workload = sys.argv[1] if len(sys.argv) > 1 else "text"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 30) - 1
check_ref = (len(sys.argv) > 3 and sys.argv[3] == "ref")
t0 = time.time()
t = gen.GENERATORS[workload](n, 3)
print(f"generated {workload} n={n} in {time.time()-t0:.1f}s", flush=True)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
d[:n] = torch.from_numpy(t).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx = M.DeviceContext(0, n)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.time()
    ctx.make_sa(d, n, sa, verbose=1 if rep == 0 else 0)
    dt = time.time() - t0
tm = ctx.timings()
print(f"SA: {dt*1e3:.1f} ms wall, {n/dt/1e6:.0f} MB/s; device {tm.total_ms:.1f} ms, rounds {tm.rounds} (doubling {tm.doubling_rounds}), unresolved after round 0: {tm.unresolved_after_round0}", flush=True)
t0 = time.time()
err = ctx.validate_sa(d, n, sa)
print(f"on-device checker: {err} errors ({time.time()-t0:.1f}s)", flush=True)
bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
t0 = time.time()
sent = ctx.bwt_from_sa(d, n, sa, bwt)
torch.cuda.synchronize()
print(f"BWT from SA: {(time.time()-t0)*1e3:.1f} ms, sentinel row {sent}, SA[sent]={int(sa[sent])}", flush=True)
inv = torch.empty(n, dtype=torch.uint8, device="cuda")
for rep in range(2):
    t0 = time.time()
    ctx.inverse_bwt(bwt, n, sent, inv)
    torch.cuda.synchronize()
    dt = time.time() - t0
ok = bool(torch.equal(inv, d[:n]))
print(f"inverse BWT: {dt*1e3:.1f} ms, {n/dt/1e6:.0f} MB/s, round trip bit-exact: {ok}", flush=True)
lcp = torch.empty(n, dtype=torch.int32, device="cuda")
t0 = time.time()
ctx.lcp(d, n, sa, lcp)
torch.cuda.synchronize()
print(f"LCP: {(time.time()-t0)*1e3:.1f} ms, max {int(lcp.max())}, mean {float(lcp.double().mean()):.2f}", flush=True)
bad = err != 0 or not ok or int(sa[sent]) != 0
if check_ref and oracle.have_reference():
    t0 = time.time()
    want = oracle.ref_make_suffix_array(t, 32)
    same = bool((sa.cpu().numpy() == want).all())
    print(f"reference (32 threads) {time.time()-t0:.1f}s: SA identical: {same}", flush=True)
    wb, ws = oracle.ref_forward_bwt(t, 32)
    same_b = bool((bwt.cpu().numpy() == wb).all()) and ws == sent
    print(f"reference BWT identical: {same_b}", flush=True)
    bad = bad or not same or not same_b
print("RESULT", "FAIL" if bad else "PASS", flush=True)
sys.exit(1 if bad else 0)
