"""CPU-only: the C-ABI library loads, exports every symbol include/msufsort_hip.h declares,
and the product path fails loudly without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from msufsort_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


def test_header_symbols_exported(L):
    from msufsort_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "msufsort_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(msufsort_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found in the header"
    assert sorted(_lib.SYMBOLS) == declared, "msufsort_amd/_lib.py SYMBOLS out of sync with the header"
    for s in declared:
        assert hasattr(L, s), f"{s} not exported"


def test_strerror(L):
    assert L.msufsort_hip_strerror(0) == b"ok"
    assert b"CPU fallback" in L.msufsort_hip_strerror(-1)


def test_no_silent_cpu_fallback(L):
    import msufsort_amd as M
    if M.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(M.MsufsortHipError):
        M.make_suffix_array(b"banana")
    with pytest.raises(M.MsufsortHipError):
        M.forward_burrows_wheeler_transform(b"banana")
    with pytest.raises(M.MsufsortHipError):
        M.DeviceContext(0)


def test_product_does_not_import_oracle():
    """The product package must never route through the oracle."""
    pkg = os.path.join(ROOT, "msufsort_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "liboracle" not in src and "libmsufsort_ref" not in src, f


def test_generators_deterministic():
    from msufsort_amd import gen
    assert gen.fnv1a64(gen.random_bytes(4096, 1)) == 0xD09EFFA23070FC72
    a = gen.text_bytes(5000, 3)
    assert a.size == 5000 and set(np.unique(a)) <= set(b"abcdefghijklmnopqrstuvwxyz \n")
    d = gen.dna_tandem_bytes(50000, 9)
    assert set(np.unique(d)) <= set(b"ACGT")


def test_bench_refuses_ranks_without_gpus():
    """`python bench.py --gpus N` starts its own ranks - and must refuse, loudly and before touching HIP, when the box has fewer
    GPUs than ranks (a silent one-GPU run is what the round-2 review found)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("MSUFSORT_BENCH_ONE_DEVICE", "WORLD_SIZE", "RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "4096", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    # a WORLD_SIZE that contradicts --gpus is an error as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_size_limits_reported_not_corrupted(L):
    """The reference corrupts its output silently from n = 2^30 on.  Here every entry point states its
    limit: the int32 rows, the inverse BWT and the LCP stop at 2^31 - 2 bytes (the forward BWT and the int64 rows go on to
    2^40 - 2 through the wide engine).  The checks sit in front of any device work, so they hold on a box without a GPU."""
    import ctypes as C
    buf = (C.c_uint8 * 16)()
    out = (C.c_int32 * 16)()
    too_large = -3
    big = (1 << 31) - 1
    assert L.msufsort_hip_make_sa_i32(buf, big, out, None) == too_large
    assert L.msufsort_hip_inverse_bwt(buf, big, 1, None) == too_large and b"2^31-2" in L.msufsort_hip_last_error()
    assert L.msufsort_hip_lcp_i32(buf, big, out, out, None) == too_large
    assert L.msufsort_hip_make_sa_i64(buf, 1 << 40, out, None) == too_large
    sent = C.c_int64(0)
    assert L.msufsort_hip_forward_bwt(buf, 1 << 40, C.byref(sent), None) == too_large
    assert L.msufsort_hip_make_sa_i32(buf, -1, out, None) == -2          # bad argument


def test_timings_struct_size_is_frozen():
    """msufsort_hip_timings is written into caller memory (timings_out): its size is part of the ABI (round-3 advisor finding;
    the header static_asserts the same number)."""
    import ctypes as C
    from msufsort_amd import _lib
    assert C.sizeof(_lib.Timings) == 216
    hdr = open(os.path.join(ROOT, "include", "msufsort_hip.h")).read()
    assert "sizeof(msufsort_hip_timings) == 216" in hdr


def test_pmc_traffic_is_keyed_by_the_library_build(L, monkeypatch):
    """bench.py quotes HBM traffic per launch from committed PMC passes (profiles/pmc_traffic.json) - only for the build of the
    library the passes ran with (round-3 review item 7): the library exports the hash of its sources, the file stores it, and a
    mismatch drops the figure with the reason instead of printing a stale one."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert isinstance(pt.get("build_id"), str) and "random" in pt["entries"] and "text" in pt["entries"]
    assert pt["entries"]["random"]["n"] == (1 << 30) - 1 and "k_partition(level 1)" in pt["entries"]["random"]["kernels"]
    assert {"k_ibwt_walk", "k_lcp", "induction (k_ind_fused + k_ind_small)"} <= set(pt["entries"]["text"]["kernels"])
    bid = L.msufsort_hip_build_id().decode()
    assert len(bid) == 16 and bench.build_id() == bid
    n = (1 << 30) - 1
    monkeypatch.setattr(bench, "build_id", lambda: pt["build_id"])
    t, src = bench.traffic_lookup("random", n, "k_partition(level 1)")
    assert t == pt["entries"]["random"]["kernels"]["k_partition(level 1)"] and "same library build" in src
    monkeypatch.setattr(bench, "build_id", lambda: "0000000000000000")
    t, src = bench.traffic_lookup("random", n, "k_partition(level 1)")
    assert t is None and src.startswith("dropped")
    assert bench.traffic_lookup("random", n + 1, "k_partition(level 1)") == (None, None)          # another size: no figure, no reason needed


@pytest.mark.parametrize("flags", [["-O2"], ["-O0", "-fsanitize=address"], ["-O2", "-D_GLIBCXX_DEBUG"]])
def test_dropin_header_compiles(L, tmp_path, flags):
    """include/library/msufsort.h against g++ -Wall -Wextra -Werror: the plain build (the result vectors grow without being
    zero-filled: MSUFSORT_UNINITIALIZED_RESIZE), and the two builds that must fall back to the value-initialising resize
    (AddressSanitizer would flag the untouched storage, the debug containers check their invariants)."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text('#include <library/msufsort.h>\n#include <cstdio>\nint main() {\n#ifdef MSUFSORT_UNINITIALIZED_RESIZE\n  std::puts("uninitialized");\n#else\n'
                   '  std::puts("value-initialised");\n#endif\n  std::vector<std::int32_t> v; maniscalco::msufsort_detail::grow_uninitialized(v, 1000); v[999] = 7; v.push_back(8);\n'
                   '  return v.size() == 1001 && v[999] == 7 && v[1000] == 8 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    libdir = os.path.join(ROOT, "msufsort_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", *flags, "-I" + os.path.join(ROOT, "include"), str(src), "-L" + libdir, "-lmsufsort_hip",
                    "-Wl,-rpath," + libdir, "-o", str(exe)], check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    assert r.stdout.strip() == ("uninitialized" if flags == ["-O2"] else "value-initialised")


REF_MAIN = "/root/reference/src/executable/msufsort/main.cpp"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree only exists in the build container")
def test_reference_consumer_compiles_unchanged(L, tmp_path):
    """The reference ships ONE consumer of its library: the demo executable, which includes <library/msufsort.h> (main.cpp:8)
    and leans on what that header includes (std::thread without <thread>, main.cpp:76).  Compiled here UNCHANGED, from where it
    lies, against this repo's include/ and linked to libmsufsort_hip.so (round-4 review: the header lacked the reference's
    transitive includes and this build failed).  oracle/Makefile keeps the same build as oracle/_ref/msufsort_demo_dropin,
    which travels to the GPU box and is RUN there (tests/test_gpu_dist.py::test_reference_demo_runs_on_the_dropin)."""
    import subprocess
    exe = tmp_path / "ref_demo"
    libdir = os.path.join(ROOT, "msufsort_amd", "lib")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), REF_MAIN, "-L" + libdir, "-lmsufsort_hip",
                        "-Wl,-rpath," + libdir, "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)          # no arguments: usage text, no device needed
    assert r.returncode == 0 and "usage: msufsort [b|s|l] input" in r.stdout
    # the include set itself, so that a box without the reference tree still notices a regression
    hdr = open(os.path.join(ROOT, "include", "library", "msufsort", "msufsort.h")).read()
    for inc in ("<vector>", "<stdint.h>", "<atomic>", "<thread>", "<memory>", "<array>", "<functional>"):
        assert "#include " + inc in hdr, inc


def test_dropin_header_keeps_the_reference_includes():
    hdr = open(os.path.join(ROOT, "include", "library", "msufsort", "msufsort.h")).read()
    for inc in ("<vector>", "<stdint.h>", "<atomic>", "<thread>", "<memory>", "<array>", "<functional>"):          # reference msufsort.h:30-36
        assert "#include " + inc in hdr, inc


def test_bench_labels_and_budget():
    """bench.py's labels must not claim what the run was not (round-4 review): the headline string only for the headline input,
    `rccl_ranks` only for an RCCL process group; and the per-rank HBM budget of BASELINE config 5 (n = 2^33 over 8 GPUs, int64
    rows) fits an MI355X while the same input on one rank's budget does not."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.metric_label("random", ["sa"], (1 << 30) - 1) == "MB/s input for SA build on 1 GiB random bytes"
    assert "1 GiB" not in bench.metric_label("random", ["sa"], 1 << 28) and "n=268435456" in bench.metric_label("random", ["sa"], 1 << 28)
    assert bench.metric_label("text", ["sa", "fbwt"], 1000) == "MB/s input for sa+fbwt on text (n=1000)"
    assert bench.rccl_ranks_of("nccl", 8) == 8 and bench.rccl_ranks_of("gloo", 8) is None
    n = 1 << 33
    b8 = bench.multi_gpu_budget(n, 8, 8, (n >> 3) + (n >> 8), False, False, False)
    assert 200 << 30 < b8["total"] < 250 << 30, b8          # ~222 GiB of the 288 GB (268 GiB) of one MI355X
    assert b8["rows"] == (n + 1) * 8 and b8["text"] == n + 64
    b2 = bench.multi_gpu_budget(n, 2, 8, n >> 1, False, False, False)
    assert b2["total"] > 288e9                               # two ranks cannot hold it: bench.py refuses with the budget in the message


def test_random_access_probe_builds_and_is_not_product_code():
    """bench.py's live random-access ceiling comes from tools/micro/window_gather.hip built as build/librandom_access_probe.so by
    __graft_entry__.build(): the symbol is there, bench.py degrades to an error entry without a GPU / without the file, and the
    product library does not reference the probe."""
    import ctypes as C
    import bench
    path = os.path.join(ROOT, "build", "librandom_access_probe.so")
    assert os.path.exists(path), "run __graft_entry__.build()"
    assert hasattr(C.CDLL(path), "msufsort_probe_random_access")
    rb = bench.request_bound({"reads_1GiB_window": 50.0}, "reads_1GiB_window", 1_000_000_000, 25.0, "x")
    assert rb["G_per_s"] == 40.0 and rb["frac_of_measured_ceiling"] == 0.8
    assert bench.request_bound({"error": "no file"}, "reads_1GiB_window", 10, 1.0, "x")["frac_of_measured_ceiling"] is None
    src = "".join(open(os.path.join(ROOT, "msufsort_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "msufsort_amd", "csrc")) if not f.endswith("Makefile"))
    assert "msufsort_probe" not in src and "window_gather" not in src
