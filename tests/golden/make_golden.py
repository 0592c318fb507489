#!/usr/bin/env python3
"""Generates tests/golden/golden.json from the UNMODIFIED reference (oracle/_ref, built
by oracle/Makefile from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

The fixture holds DATA only: inputs (literal or generator+seed+size), and the expected
outputs of the reference (literal arrays for tiny cases; FNV-1a-64 hashes + probes for
large ones).  Convention notes: SURVEY.md section 4.3 / 8(c).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import oracle  # noqa: E402
from msufsort_amd import gen  # noqa: E402


sweep_input = gen.sweep_bytes


def case(text: np.ndarray, threads=1, literal=False):
    sa = oracle.ref_make_suffix_array(text, threads)
    bwt, sent = oracle.ref_forward_bwt(text, threads)
    inv = oracle.ref_reverse_bwt(bwt, sent, threads)
    assert (inv == text).all()
    lcp = oracle.ref_lcp(text, sa, 1)
    d = {
        "n": int(text.size),
        "input_fnv": "%016x" % oracle.fnv1a64(text),
        "sa_fnv": "%016x" % oracle.fnv1a64(sa),
        "sa_first": int(sa[1]), "sa_last": int(sa[-1]),
        "bwt_fnv": "%016x" % oracle.fnv1a64(bwt),
        "sentinel": int(sent),
        "lcp_fnv": "%016x" % oracle.fnv1a64(lcp),
    }
    if literal:
        d["sa"] = sa.tolist()
        d["bwt"] = bwt.tolist()
        d["lcp"] = lcp.tolist()
    return d


def main():
    assert oracle.have_reference(), "build oracle/_ref first (make -C oracle)"
    out = {"literal": [], "generated": [], "sweep": []}
    lits = [b"\x07", b"\x01\x02", b"\x02\x01", b"\x05\x05", b"\x03\x01\x02", b"banana", b"mississippi",
            b"A" * 40, b"ACGT" * 75, b"ab" * 500, b"\x00", b"\x00" * 17, b"ab\x00\x00", b"\x00\x00ab\x00\x00\x00",
            b"abracadabra" * 3, bytes(range(256)), bytes(range(255, -1, -1)), b"\xff" * 33, b"a\x00" * 20]
    for s in lits:
        t = np.frombuffer(s, dtype=np.uint8)
        d = case(t, literal=True)
        d["text"] = list(s)
        out["literal"].append(d)
    gens = [("random", 1, 4096), ("random", 12345, 1 << 20), ("random", 12345, (1 << 20) + 3),
            ("dna", 7, 1 << 20), ("dna_tandem", 9, 300000), ("text", 3, 1 << 20),
            ("random", 5, 70000), ("text", 11, 200000), ("dna", 21, 65537)]
    for name, seed, n in gens:
        t = gen.GENERATORS[name](n, seed)
        d = case(t, threads=4)
        d.update({"generator": name, "seed": seed})
        out["generated"].append(d)
    # 37-mer repeated x5000 (SURVEY 4.3 last row)
    u = gen.dna_bytes(37, 9)
    t = np.tile(u, 5000)
    d = case(t, threads=4)
    d.update({"generator": "tile37", "seed": 9})
    out["generated"].append(d)
    # a thinned version of the demo's self-test sweep (main.cpp:389-435)
    for a in (1, 2, 3, 4, 7, 16, 64, 255):
        for n in (1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 511, 1023):
            t = sweep_input(a, n)
            d = case(t)
            d.update({"alphabet": a})
            out["sweep"].append(d)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
