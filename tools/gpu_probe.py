#!/usr/bin/env python3
"""GPU-box diagnostic: runs groups of SA/BWT/LCP cases through the C-ABI and compares with the
oracle (port and, when present, the reference .so).  Never stops at the first failure.

    python tools/gpu_probe.py <group> [...]      groups: lit sweep gen big hist
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import msufsort_amd as M  # noqa: E402
import oracle  # noqa: E402
from msufsort_amd import gen  # noqa: E402

fails = 0


def report(name, ok, extra=""):
    global fails
    if not ok:
        fails += 1
    print(("PASS " if ok else "FAIL ") + name + (" " + extra if extra else ""), flush=True)


def first_diff(a, b):
    d = np.nonzero(a != b)[0]
    return "" if d.size == 0 else f"first diff at {int(d[0])}: got {a[d[0]:d[0]+6].tolist()} want {b[d[0]:d[0]+6].tolist()} ndiff={d.size}"


def check_sa(name, t, want=None, verbose=0, text_rounds=0):
    t = np.ascontiguousarray(t, dtype=np.uint8)
    try:
        t0 = time.time()
        sa = M.make_suffix_array(t, verbose=verbose, text_rounds=text_rounds)
        dt = time.time() - t0
    except Exception as e:  # noqa: BLE001
        report(name, False, f"exception {e}")
        return None
    if want is None:
        want = oracle.ref_make_suffix_array(t, 8) if (oracle.have_reference() and t.size) else oracle.make_suffix_array(t)
    ok = sa.shape == want.shape and bool((sa == want).all())
    report(name, ok, f"n={t.size} {dt*1e3:.1f} ms " + ("" if ok else first_diff(sa, want)))
    return sa


def g_lit():
    g = json.load(open(os.path.join(ROOT, "tests/golden/golden.json")))
    for d in g["literal"]:
        t = np.array(d["text"], dtype=np.uint8)
        check_sa("lit:" + bytes(d["text"][:10]).hex(), t, np.array(d["sa"], dtype=np.int32))
    check_sa("lit:empty", np.zeros(0, dtype=np.uint8), np.array([0], dtype=np.int32))


def g_sweep():
    bad = 0
    cnt = 0
    for a in (1, 2, 3, 4, 7, 16, 64, 255):
        for n in (1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 511, 1023, 4097, 20000):
            t = gen.sweep_bytes(a, n)
            sa = M.make_suffix_array(t)
            want = oracle.make_suffix_array(t) if a > 2 or n < 2000 else oracle.ref_make_suffix_array(t)
            cnt += 1
            if not (sa == want).all():
                bad += 1
                if bad <= 10:
                    report(f"sweep a={a} n={n}", False, first_diff(sa, want))
    report(f"sweep {cnt} cases", bad == 0, f"bad={bad}")


def g_gen():
    for name, seed, n in [("random", 1, 4096), ("random", 5, 70000), ("random", 12345, 1 << 20), ("random", 12345, (1 << 20) + 3),
                          ("dna", 21, 65537), ("dna", 7, 1 << 20), ("text", 11, 200000), ("text", 3, 1 << 20),
                          ("dna_tandem", 9, 300000)]:
        check_sa(f"gen:{name}:{seed}:{n}", gen.GENERATORS[name](n, seed), verbose=1)
    check_sa("gen:tile37", np.tile(gen.dna_bytes(37, 9), 5000), verbose=1)
    check_sa("gen:allA", np.full(100000, 65, dtype=np.uint8), verbose=1)
    check_sa("gen:allzero", np.zeros(5000, dtype=np.uint8), verbose=1)
    t = gen.random_bytes(50000, 3); t[-3000:] = 0
    check_sa("gen:trailzeros", t, verbose=1)
    t = gen.random_bytes(300000, 4) % 2
    check_sa("gen:binary", t.astype(np.uint8), verbose=1)
    check_sa("gen:text-doubling-early", gen.text_bytes(300000, 5), verbose=1, text_rounds=1)


def g_big():
    for name, seed, n in [("random", 12345, 1 << 24), ("text", 3, 1 << 24), ("dna", 7, 1 << 24), ("random", 12345, 1 << 26)]:
        check_sa(f"big:{name}:{n}", gen.GENERATORS[name](n, seed), verbose=1)


def g_hist():
    import torch
    ctx = M.DeviceContext(0)
    for n in (1, 17, 4096, 100001, 1 << 22):
        t = gen.random_bytes(n, 77)
        d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        d[:n] = torch.from_numpy(t).cuda()
        h = torch.zeros(65536, dtype=torch.int32, device="cuda")
        ctx.debug_hist16(d, n, h)
        tt = np.concatenate([t, np.zeros(1, np.uint8)]).astype(np.uint32)
        want = np.bincount((tt[:-1] << 8) | tt[1:], minlength=65536)
        got = h.cpu().numpy()
        report(f"hist16 n={n}", bool((got == want).all()), first_diff(got, want))


if __name__ == "__main__":
    oracle.build()
    print("devices:", M.device_count(), "reference .so:", oracle.have_reference(), flush=True)
    for g in sys.argv[1:]:
        t0 = time.time()
        try:
            globals()["g_" + g]()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            report("group " + g, False, str(e))
        print(f"-- group {g} done in {time.time()-t0:.1f}s", flush=True)
    print("TOTAL FAILS", fails, flush=True)
    sys.exit(1 if fails else 0)
