"""CPU-only: pins the plain-C restatement (oracle/msufsort_oracle.c) against the golden
vectors produced by the unmodified reference (tests/golden/make_golden.py) and, where the
reference .so is present, against the live reference."""
import numpy as np
import pytest

from msufsort_amd import gen


def _input(d):
    if "text" in d:
        return np.array(d["text"], dtype=np.uint8)
    if "alphabet" in d:
        return gen.sweep_bytes(d["alphabet"], d["n"])
    if d["generator"] == "tile37":
        return np.tile(gen.dna_bytes(37, d["seed"]), 5000)
    return gen.GENERATORS[d["generator"]](d["n"], d["seed"])


def _check(oracle, d, literal=False):
    t = _input(d)
    assert t.size == d["n"]
    assert "%016x" % oracle.fnv1a64(t) == d["input_fnv"], "generator drifted"
    sa = oracle.make_suffix_array(t)
    assert "%016x" % oracle.fnv1a64(sa) == d["sa_fnv"]
    assert sa[0] == d["n"] and sa[1] == d["sa_first"] and sa[-1] == d["sa_last"]
    bwt, sent = oracle.forward_bwt(t)
    assert sent == d["sentinel"]
    assert "%016x" % oracle.fnv1a64(bwt) == d["bwt_fnv"]
    assert (oracle.reverse_bwt(bwt, sent) == t).all()
    lcp = oracle.lcp(t, sa)
    assert "%016x" % oracle.fnv1a64(lcp) == d["lcp_fnv"]
    if literal:
        assert sa.tolist() == d["sa"]
        assert bwt.tolist() == d["bwt"]
        assert lcp.tolist() == d["lcp"]
    assert oracle.validate_sa(t, sa) == 0


def test_known_answers_survey(oracle_mod):
    # SURVEY.md section 4.3 (probed from the reference)
    o = oracle_mod
    assert o.make_suffix_array(b"banana").tolist() == [6, 5, 3, 1, 0, 4, 2]
    assert o.make_suffix_array(b"mississippi").tolist() == [11, 10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2]
    b, s = o.forward_bwt(b"banana")
    assert bytes(b) == b"annbaa" and s == 4
    b, s = o.forward_bwt(b"mississippi")
    assert bytes(b) == b"ipssmpissii" and s == 5
    assert o.lcp(b"banana", o.make_suffix_array(b"banana")).tolist()[:5] == [1, 3, 0, 0, 2]
    x = gen.random_bytes(4096, 1)
    assert o.fnv1a64(x) == 0xD09EFFA23070FC72 == gen.fnv1a64(x)
    assert o.fnv1a64(o.make_suffix_array(x)) == 0xAA58C5B9184D33C9


def test_empty_input(oracle_mod):
    assert oracle_mod.make_suffix_array(b"").tolist() == [0]


def test_literal_golden(oracle_mod, golden):
    for d in golden["literal"]:
        _check(oracle_mod, d, literal=True)


def test_sweep_golden(oracle_mod, golden):
    for d in golden["sweep"]:
        _check(oracle_mod, d)


@pytest.mark.parametrize("i", range(10))
def test_generated_golden(oracle_mod, golden, i):
    d = golden["generated"][i]
    if d["generator"] in ("tile37", "dna_tandem"):
        pytest.skip("periodic input: the port omits the tandem-repeat shortcut (slow); covered by _ref")
    _check(oracle_mod, d)


def test_port_vs_live_reference(oracle_mod):
    o = oracle_mod
    if not o.have_reference():
        pytest.skip("oracle/_ref not built here")
    for a in (1, 2, 4, 26, 256):
        for n in (1, 2, 7, 63, 1000, 5000):
            t = gen.sweep_bytes(a, n) if a < 256 else gen.random_bytes(n, n)
            assert (o.make_suffix_array(t) == o.ref_make_suffix_array(t)).all()
            b1, s1 = o.forward_bwt(t)
            b2, s2 = o.ref_forward_bwt(t)
            assert s1 == s2 and (b1 == b2).all()
            assert (o.ref_reverse_bwt(b1, s1) == t).all()
            sa = o.make_suffix_array(t)
            assert (o.lcp(t, sa) == o.ref_lcp(t, sa)).all()
