#!/usr/bin/env python3
"""Full-size golden vectors (BASELINE configs 3/4 and the config-5 stream) from the UNMODIFIED reference
(oracle/_ref, built by oracle/Makefile from /root/reference).  Build container only; takes tens of minutes and
~20 GB of RAM on 8 cores:

    python tests/golden/make_golden_full.py            # writes tests/golden/golden_full.json

Data only: generator + seed + size of every input and FNV-1a-64 hashes / probes of what the reference returned.
The GPU box compares its HBM-resident results against these hashes (tests/test_gpu_full.py) - no reference there.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import oracle  # noqa: E402
from msufsort_amd import gen  # noqa: E402

CASES = [
    # name, seed, n, what
    ("text", 3, (1 << 30) - 1, ("sa", "bwt", "lcp")),            # config 3 / 4
    ("dna", 2024, (1 << 30) - 1, ("sa", "bwt")),                 # first 2^30-1 bytes of the config-5 style stream of tests/test_gpu_big.py
    ("dna_tandem", 9, 1 << 28, ("sa", "bwt")),                   # config-5 workload (long tandem repeats) at 256 MiB
    ("random", 12345, (1 << 30) - 1, ("sa", "bwt")),             # the bench headline (SURVEY 8(d) north-star input)
    ("random", 12345, 1 << 28, ("sa",)),                         # config 2
]


def main():
    assert oracle.have_reference(), "build oracle/_ref first (make -C oracle)"
    threads = os.cpu_count() or 1
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_full.json")
    out = {"full": []}
    only = sys.argv[1:]
    if only and os.path.exists(path):
        out = json.load(open(path))
    for name, seed, n, what in CASES:
        if only and name not in only and f"{name}:{n}" not in only:
            continue
        t0 = time.time()
        t = gen.GENERATORS[name](n, seed)
        d = {"generator": name, "seed": seed, "n": n, "input_fnv": "%016x" % oracle.fnv1a64(t)}
        sa = oracle.ref_make_suffix_array(t, threads)
        d.update({"sa_fnv": "%016x" % oracle.fnv1a64(sa), "sa_first": int(sa[1]), "sa_last": int(sa[-1]),
                  "sa_probe": [int(sa[i]) for i in (2, n // 3, n // 2, n - 7)]})
        print(name, n, "SA", round(time.time() - t0, 1), "s", flush=True)
        if "lcp" in what:
            lcp = oracle.ref_lcp(t, sa, threads)
            d["lcp_fnv"] = "%016x" % oracle.fnv1a64(lcp)
            del lcp
            print(name, n, "LCP", round(time.time() - t0, 1), "s", flush=True)
        del sa
        if "bwt" in what:
            bwt, sent = oracle.ref_forward_bwt(t, threads)
            d.update({"bwt_fnv": "%016x" % oracle.fnv1a64(bwt), "sentinel": int(sent)})
            del bwt
            print(name, n, "BWT", round(time.time() - t0, 1), "s", flush=True)
        out["full"] = [x for x in out["full"] if not (x["generator"] == name and x["n"] == n)] + [d]
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
