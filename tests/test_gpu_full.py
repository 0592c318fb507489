"""BASELINE configs 3 / 4 (and the config-5 style streams) at FULL size on the GPU box, against hashes the unmodified
reference produced in the build container (tests/golden/golden_full.json, made by tests/golden/make_golden_full.py):
SA + forward BWT + inverse BWT + LCP, everything resident in HBM; nothing here needs the reference at run time."""
import json
import os

import numpy as np
import pytest

from msufsort_amd import gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "golden", "golden_full.json")
CASES = json.load(open(PATH))["full"] if os.path.exists(PATH) else []


def _fnv(oracle, t):
    return "%016x" % oracle.fnv1a64(t.cpu().numpy())


@pytest.mark.parametrize("d", CASES, ids=[f"{c['generator']}-{c['n']}" for c in CASES])
def test_full_size_golden(oracle_mod, d):
    import torch

    import msufsort_amd as M
    n = d["n"]
    t = gen.GENERATORS[d["generator"]](n, d["seed"])
    assert "%016x" % oracle_mod.fnv1a64(t) == d["input_fnv"]
    dev = torch.device("cuda")
    ctx = M.DeviceContext(0)
    dt = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    dt[:n] = torch.from_numpy(t).to(dev)
    sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ctx.make_sa(dt, n, sa)
    assert int(sa[0]) == n and int(sa[1]) == d["sa_first"] and int(sa[n]) == d["sa_last"]
    assert [int(sa[i]) for i in (2, n // 3, n // 2, n - 7)] == d["sa_probe"]
    assert _fnv(oracle_mod, sa) == d["sa_fnv"], "suffix array differs from the reference"
    if "bwt_fnv" in d:
        bwt = torch.empty(n, dtype=torch.uint8, device=dev)
        sent = ctx.bwt_from_sa(dt, n, sa, bwt)
        assert sent == d["sentinel"] and _fnv(oracle_mod, bwt) == d["bwt_fnv"], "BWT differs from the reference"
        inv = torch.empty(n, dtype=torch.uint8, device=dev)
        ctx.inverse_bwt(bwt, n, sent, inv)
        assert torch.equal(inv, dt[:n]), "inverse BWT does not restore the text"
        # the forward transform as one call (two-stage builds take its bytes from the rows' preceding characters)
        sent2 = ctx.forward_bwt(dt, n, inv)
        assert sent2 == d["sentinel"] and _fnv(oracle_mod, inv) == d["bwt_fnv"], "forward BWT (one call) differs from the reference"
        del bwt, inv
    if "lcp_fnv" in d:
        lcp = torch.empty(n, dtype=torch.int32, device=dev)
        ctx.lcp(dt, n, sa, lcp)
        assert _fnv(oracle_mod, lcp) == d["lcp_fnv"], "LCP differs from the reference"
        del lcp
    if d["generator"] == "dna":
        # the same rows from the WIDE engine (what runs beyond 2^31 - 2 bytes): 40-bit indices, 4 logical shards
        ctx.trim()
        sa64 = torch.empty(n + 1, dtype=torch.int64, device=dev)
        ctx.make_sa_i64(dt, n, sa64, force_wide=True, n_shards=4)
        assert bool((sa64 == sa).all()), "wide engine differs from the narrow one"
        del sa64
    del sa, dt
    torch.cuda.empty_cache()
