#!/usr/bin/env python3
"""Round 6: host-pointer calls on inputs of many shapes - alphabets of 3 .. 200 byte values with flat and skewed frequencies, runs, sizes
around the ring and policy thresholds - through the default policy and with the two-stage build forced: every row of the suffix array
and every byte + the sentinel row of the forward transform against a device-resident sort-all build.  usage: gpu_host_fuzz.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import msufsort_amd as M
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
bad = 0
for case in range(cases):
    sigma = int(rng.choice([3, 4, 5, 8, 16, 17, 29, 64, 96, 128, 200]))
    n = int(rng.integers(60 << 20, 180 << 20)) | 1
    syms = np.sort(rng.choice(np.arange(1, 256), size=sigma, replace=False)).astype(np.uint8)
    skew = float(rng.choice([0.0, 1.0, 2.5]))
    p = 1.0 / np.arange(1, sigma + 1) ** skew; p /= p.sum()
    t = syms[rng.choice(sigma, size=n, p=p).astype(np.int64)]
    if rng.random() < 0.5:
        a = int(rng.integers(0, n - 5000)); t[a:a + int(rng.integers(2, 4000))] = syms[int(rng.integers(0, sigma))]
    t = np.ascontiguousarray(t)
    ctx = M.DeviceContext(0)
    d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
    ref = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ctx.make_sa(d, n, ref, two_stage=-1)
    errs = ctx.validate_sa(d, n, ref)
    bref = torch.empty(n, dtype=torch.uint8, device="cuda")
    sref = ctx.bwt_from_sa(d, n, ref, bref)
    want, bwant = ref.cpu().numpy(), bref.cpu().numpy()
    del ctx, d, ref, bref
    torch.cuda.empty_cache()
    line = f"case {case}: sigma {sigma} skew {skew} n {n} ({n >> 20} MiB) checker errors {errs}:"
    for ts in (0, 1):
        t0 = time.perf_counter(); sa = M.make_suffix_array(t, two_stage=ts); t1 = time.perf_counter()
        b, s = M.forward_burrows_wheeler_transform(t, two_stage=ts); t2 = time.perf_counter()
        ok = bool((sa == want).all()) and s == sref and bool((b == bwant).all())
        bad += (not ok) or errs != 0
        line += f"  two_stage={ts}: sa {1e3 * (t1 - t0):.0f} ms fbwt {1e3 * (t2 - t1):.0f} ms {'ok' if ok else 'MISMATCH'}"
        del sa, b
    print(line, flush=True)
print("RESULT", "PASS" if bad == 0 else f"FAIL ({bad})")
