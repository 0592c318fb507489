#!/usr/bin/env python3
"""bench.py - MB/s of input for the suffix-array build on uniform-random bytes (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--size BYTES] [--workload random|text|dna|dna_tandem] [--op sa[,bwt][,fbwt][,ibwt][,lcp]]

Default (what the driver runs): the headline, SA of 2^30 - 1 uniform random bytes (splitmix64 seed 12345), checked after the timed
region against the hash the UNMODIFIED reference produced for the same input (tests/golden/golden_full.json).  With N = 1 and
no workload flags the same JSON line also carries BASELINE configs 3 and 4 under "configs" (text, 2^30 - 1 bytes: SA, forward
BWT, inverse BWT, LCP), measured after the headline, each with its own roofline entry, reference hashes and CPU baseline.

A "step" is one complete suffix-array build (16-bit radix histogram, two 8-bit scatter levels, LDS bucket
sorts, refinement rounds) of one synthetic input that is already resident in HBM.  N = 1: the whole
array on one MI355X.  N > 1: the 4-byte-prefix space is split into N count-balanced ranges, every rank sorts its range into
its slice of the full array and the slices are exchanged with one all-gatherv (one group of direct sends/receives over
RCCL/xGMI) that is complete on every rank before the step ends - `value` is the rate of whole builds one after the other
(latency), the pipelined rate (build k+1 under the exchange of build k) a secondary field; total work is fixed, so "scaling"
is "strong".  `python bench.py --gpus N` starts its N ranks itself (a child
`torch.distributed.run`, one rank per GPU); started under torch.distributed.run it is one of the ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
CPU_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import oracle
from msufsort_amd import gen
t = gen.GENERATORS[%(workload)r](%(n)d, %(seed)d)
out = []
bwt = None
for threads, what in %(runs)r:
    if what == "ibwt" and bwt is None:
        bwt = oracle.ref_forward_bwt(t, 16)          # (input of the inverse: outside the timed part)
    t0 = time.perf_counter()
    if what == "sa":
        oracle.ref_make_suffix_array(t, threads)
    elif what == "ibwt":
        oracle.ref_reverse_bwt(bwt[0], bwt[1], threads)
    else:
        oracle.ref_forward_bwt(t, threads)
    dt = time.perf_counter() - t0
    out.append({"threads": threads, "op": what, "seconds": round(dt, 3), "MB/s": round(t.size / dt / 1e6, 2)})
    print(json.dumps(out), flush=True)
"""


def cpu_reference_runs(workload, seed, n, runs, timeout_s):
    """The unmodified reference (oracle/_ref) timed on the host cores in a CHILD process (its spin-wait worker pool can stall;
    a watchdog ends it).  Checker/baseline only - never the product.  Returns the runs that finished."""
    import subprocess
    code = CPU_CHILD % {"root": ROOT, "workload": workload, "n": n, "seed": seed, "runs": runs}
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("[")]
    except subprocess.TimeoutExpired as e:
        so = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        lines = [ln for ln in so.splitlines() if ln.startswith("[")]
    return json.loads(lines[-1]) if lines else []


def cpu_baseline(sample_bytes, seed, workload="random", op="sa", full_n=None, all_threads=False, median_of=1, t1_prefix=0, counts=None):
    """cpu_baseline object of the JSON line: the unmodified reference ("reference") on the host cores, or - when oracle/_ref is
    absent - our single-threaded C restatement ("port") on a bounded sample.
    SURVEY 8(d) protocol (median_of = 3, the headline): one run with 16 and one with 32 threads, then the faster count again until it
    has `median_of` runs - `value` is the MEDIAN of those (the reference's spin-wait pool makes single draws noisy) -, plus ONE
    single-thread run on the first `t1_prefix` bytes of the same stream (labelled; T = 1 on the whole GiB would take minutes)."""
    import oracle
    ncpu = os.cpu_count() or 1
    if oracle.have_reference():
        # The reference's workers spin-wait and its serial phases do not shrink: on the 2 x 128-thread host of the GPU box it is
        # FASTEST with about 16 threads (256 MiB: 146 MB/s with 16, 80 with 32, 26 with 64, 13 with 128, no end within 90 s with
        # 192 or 256: profiles/r03_cpu_reference_threads.txt).
        counts = sorted(set(counts or [max(1, min(16, ncpu)), max(1, min(32, ncpu))]))
        res = cpu_reference_runs(workload, seed, sample_bytes, [(c_, op) for c_ in counts], timeout_s=420)
        if not res:
            return {"error": "the reference did not finish within 420 s", "host_cpus": ncpu}
        best = max(res, key=lambda r: r["MB/s"])
        if median_of > 1:
            more = cpu_reference_runs(workload, seed, sample_bytes, [(best["threads"], op)] * (median_of - 1), timeout_s=420)
            res = res + more
        at_best = sorted(r["MB/s"] for r in res if r["threads"] == best["threads"])
        value = at_best[len(at_best) // 2] if len(at_best) % 2 else round((at_best[len(at_best) // 2 - 1] + at_best[len(at_best) // 2]) / 2, 2)
        what = {"sa": "make_suffix_array", "ibwt": "reverse_burrows_wheeler_transform"}.get(op, "forward_burrows_wheeler_transform")
        part = "the whole input" if full_n == sample_bytes else f"first {sample_bytes} bytes of the same stream"
        how = (f"one run per thread count ({', '.join(str(c_) for c_ in counts)}), then the faster count repeated: value = median of its {len(at_best)} runs"
               if median_of > 1 else f"one run per thread count ({', '.join(str(r['threads']) for r in res)}); value = the faster")
        out = {"value": value, "unit": "MB/s", "cores": best["threads"], "kind": "reference", "host_cpus": ncpu, "runs": res,
               "runs_at_reported_threads": len(at_best), "sample": f"{part}, {what} wall time incl. allocation (main.cpp:437-453), {how}"}
        if t1_prefix:
            small = min(sample_bytes, t1_prefix)
            r1 = cpu_reference_runs(workload, seed, small, [(1, op)], timeout_s=240)
            out["single_thread"] = (dict(r1[0], sample_bytes=small, sample=f"first {small} bytes of the same stream, one run") if r1 else
                                    {"threads": 1, "sample_bytes": small, "error": "did not finish within 240 s"})
        if all_threads and ncpu > max(counts):
            small = min(sample_bytes, 1 << 25)
            r2 = cpu_reference_runs(workload, seed, small, [(ncpu, op)], timeout_s=45)
            out["all_hardware_threads"] = (dict(r2[0], sample_bytes=small) if r2 else
                                           {"threads": ncpu, "sample_bytes": small, "error": "did not finish within 45 s (spin-wait worker pool)"})
        return out
    from msufsort_amd import gen
    t = gen.GENERATORS[workload](min(sample_bytes, 1 << 24), seed)
    t0 = time.perf_counter()
    oracle.make_suffix_array(t)
    dt = time.perf_counter() - t0
    return {"value": round(t.size / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port", "host_cpus": ncpu,
            "sample": f"first {t.size} bytes of the same stream, C restatement of the two-stage sort, one run"}


def golden_entry(workload, seed, n):
    try:
        full = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json")))["full"]
    except Exception:  # noqa: BLE001
        return None
    for e in full:
        if e["generator"] == workload and e["seed"] == seed and e["n"] == n:
            return e
    return None


def check_golden(entry, d_sa=None, d_bwt=None, sentinel=None, d_lcp=None):
    """After the timed region: FNV-1a-64 of the device results (copied to the host) against what the unmodified reference
    returned for the same input in the build container (tests/golden/make_golden_full.py).  The hash routine is the checker's."""
    import oracle
    ok, checked = True, []
    if d_sa is not None and "sa_fnv" in entry:
        ok = ok and ("%016x" % oracle.fnv1a64(d_sa.cpu().numpy())) == entry["sa_fnv"]; checked.append("sa")
    if d_bwt is not None and "bwt_fnv" in entry:
        ok = ok and ("%016x" % oracle.fnv1a64(d_bwt.cpu().numpy())) == entry["bwt_fnv"] and int(sentinel) == entry["sentinel"]; checked.append("bwt")
    if d_lcp is not None and "lcp_fnv" in entry:
        ok = ok and ("%016x" % oracle.fnv1a64(d_lcp.cpu().numpy())) == entry["lcp_fnv"]; checked.append("lcp")
    return ok, checked


def pcie_floor(torch, dev, n):
    """What the PCIe link alone needs for the drop-in's buffers, measured here with pinned memory: the text (n bytes) to the
    device, the suffix array (4(n+1) bytes) back.  Nothing else is in these numbers - no page faults, no sort."""
    import time as _t
    h_text = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    h_sa = torch.empty(n + 1, dtype=torch.int32, pin_memory=True)
    h_text.zero_(); h_sa.zero_()
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    d_sa = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    best = [1e9, 1e9]
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = _t.perf_counter(); d_text.copy_(h_text, non_blocking=True); torch.cuda.synchronize(dev); t1 = _t.perf_counter()
        h_sa.copy_(d_sa, non_blocking=True); torch.cuda.synchronize(dev); t2 = _t.perf_counter()
        best = [min(best[0], (t1 - t0) * 1e3), min(best[1], (t2 - t1) * 1e3)]
    del h_text, h_sa, d_text, d_sa
    return {"h2d_text_ms": round(best[0], 2), "d2h_sa_ms": round(best[1], 2), "h2d_GBps": round(n / best[0] / 1e6, 1), "d2h_GBps": round(4 * (n + 1) / best[1] / 1e6, 1),
            "sa_floor_ms": round(best[0] + best[1], 2), "bwt_floor_ms": round(best[0] * 2, 2),
            "note": "pinned-memory copies measured in this run: text to the device + result back (SA: 4(n+1) bytes; BWT / inverse: n bytes each way)"}


def host_bench_path():
    """build/host_bench (made by __graft_entry__.build()); compiled here when it did not travel with the tree."""
    exe = os.path.join(ROOT, "build", "host_bench")
    if not os.path.exists(exe):
        import subprocess
        try:
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "host_bench.cpp"),
                            "-L" + os.path.join(ROOT, "msufsort_amd", "lib"), "-lmsufsort_hip", "-Wl,-rpath,$ORIGIN/../msufsort_amd/lib", "-o", exe],
                           check=True, capture_output=True, timeout=120)
        except Exception:  # noqa: BLE001
            pass
    return exe


CABI_CHILD = r"""
import ctypes as C, json, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from msufsort_amd import _lib
from msufsort_amd.api import _opts
t = np.fromfile(%(path)r, dtype=np.uint8)
n = t.size
L = _lib.lib()
ms, chk, h = [], None, None
for r in range(%(reps)d + 1):
    sa = np.empty(n + 1, dtype=np.int32)                  # fresh, untouched memory every time
    src = t.copy()                                        # ... and a text buffer the runtime has not met before
    o = _opts(n_shards=0)                                 # (0: the entry point's own choice - 8 streamed key-range shards per device)
    dv = (C.c_int32 * 1)(%(device)d)
    t0 = time.perf_counter()
    _lib.check(L.msufsort_hip_make_sa_multi(dv, 1, src.ctypes.data, n, sa.ctypes.data, 4, C.byref(o), None), "make_sa_multi")
    dt = (time.perf_counter() - t0) * 1e3
    if r:
        ms.append(round(dt, 2))
    else:
        first = round(dt, 2)
    chk = (int(sa[0]), int(sa[1]), int(sa[n]))
    if r == %(reps)d and %(hash)d:
        import oracle                                     # the checker's hash routine, after the timed calls
        h = "%%016x" %% oracle.fnv1a64(sa)
    del sa
print(json.dumps({"sa_ms": ms, "first_call_ms": first, "sa_0_1_n": chk, "sa_fnv": h}))
"""


def end_to_end_host(torch, dev, workload, seed, n, floor, golden=None, reps=3):
    """The drop-in as a caller sees it (reference main.cpp:386,440-442): pageable host
    text in, FRESHLY allocated host result out, clock around the call.  Two legs, each in a process of its own (a caller's
    process, not this one with its torch allocator and worker threads): the C-ABI entry point through ctypes
    (msufsort_hip_make_sa_multi into np.empty) and the C++ header (examples/host_bench.cpp: maniscalco::msufsort::
    make_suffix_array incl. the construction of its std::vector, forward and inverse transform in place)."""
    import subprocess
    from msufsort_amd import gen
    out = {"workload": f"{workload} (seed {seed}), n={n}", "pcie_floor": floor}
    path = f"/dev/shm/msufsort_bench_{os.getpid()}_{workload}.bin"
    try:
        gen.GENERATORS[workload](n, seed).tofile(path)
        code = CABI_CHILD % {"root": ROOT, "path": path, "reps": reps, "device": dev.index or 0, "hash": 1 if (golden is not None and "sa_fnv" in golden) else 0}
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            ca = json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-300:]}
        except Exception as e:  # noqa: BLE001
            ca = {"error": repr(e)}
        if ca.get("sa_ms"):
            ca["sa_best_ms"] = min(ca["sa_ms"]); ca["sa_MBps"] = round(n / min(ca["sa_ms"]) / 1e3, 1)
            ca["ratio_to_pcie_floor"] = round(min(ca["sa_ms"]) / floor["sa_floor_ms"], 3)
            if golden is not None and "sa_fnv" in golden:
                ca["valid"] = ca.get("sa_fnv") == golden["sa_fnv"]
        ca["entry"] = "msufsort_hip_make_sa_multi (ctypes, own process; numpy pageable text -> np.empty(n+1))"
        out["c_abi"] = ca
        exe = host_bench_path()
        if os.path.exists(exe):
            try:
                r = subprocess.run([exe, path, str(reps)], capture_output=True, text=True, timeout=600)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                hb = json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-300:]}
            except Exception as e:  # noqa: BLE001
                hb = {"error": repr(e)}
            if "sa_ms" in hb and hb["sa_ms"]:
                hb["sa_best_ms"] = min(hb["sa_ms"]); hb["sa_MBps"] = round(n / min(hb["sa_ms"]) / 1e3, 1)
                hb["sa_ratio_to_pcie_floor"] = round(min(hb["sa_ms"]) / floor["sa_floor_ms"], 3)
                hb["forward_bwt_best_ms"] = min(hb["forward_bwt_ms"]); hb["inverse_bwt_best_ms"] = min(hb["inverse_bwt_ms"])
                hb["forward_bwt_MBps"] = round(n / min(hb["forward_bwt_ms"]) / 1e3, 1); hb["inverse_bwt_MBps"] = round(n / min(hb["inverse_bwt_ms"]) / 1e3, 1)
                if golden is not None and "sa_fnv" in golden:
                    hb["valid"] = hb.get("sa_fnv") == golden["sa_fnv"] and bool(hb.get("round_trip")) and (golden.get("sentinel") in (None, hb.get("sentinel")))
            hb["entry"] = "include/library/msufsort.h: maniscalco::msufsort::make_suffix_array (result vector constructed inside the timed call), forward_burrows_wheeler_transform, reverse_burrows_wheeler_transform; examples/host_bench.cpp"
            out["cpp_header"] = hb
        else:
            out["cpp_header"] = {"error": "build/host_bench is missing (python -c 'import __graft_entry__ as g; g.build()')"}
    finally:
        if os.path.exists(path):
            os.remove(path)
    return out


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` run directly: start the N ranks as a CHILD (torch.distributed.run, one rank per GPU) and
    hand its exit code back - the library's own fan-out over its workers is msufsort.cpp:1652-1683.  Nothing in this
    process has touched HIP yet (device_count() does not initialise the GPU), and nothing is exec'ed."""
    import socket
    import subprocess

    import torch
    have = torch.cuda.device_count()
    if have < n_ranks and not os.environ.get("MSUFSORT_BENCH_ONE_DEVICE"):
        print(f"bench.py: --gpus {n_ranks} but only {have} GPU(s) visible "
              "(MSUFSORT_BENCH_ONE_DEVICE=1 lets all ranks share GPU 0 over gloo: a test hook, not a measurement)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


class Single:
    """One single-GPU measurement: `steps` timed steps of ops over one input resident in HBM."""

    def __init__(self, M, torch, ctx, dev, workload, seed, n, ops):
        from msufsort_amd import gen
        self.M, self.torch, self.ctx, self.dev, self.workload, self.seed, self.n, self.ops = M, torch, ctx, dev, workload, seed, n, ops
        t = gen.GENERATORS[workload](n, seed)
        self.d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        self.d_text[:n] = torch.from_numpy(t).to(dev)
        del t
        self.d_sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
        need_bwt = any(o in ops for o in ("bwt", "ibwt", "fbwt"))
        self.d_bwt = torch.empty(n, dtype=torch.uint8, device=dev) if need_bwt else None
        self.d_inv = torch.empty(n, dtype=torch.uint8, device=dev) if "ibwt" in ops else None
        self.d_lcp = torch.empty(n, dtype=torch.int32, device=dev) if "lcp" in ops else None
        torch.cuda.synchronize(dev)          # the engine works on its own HIP stream: the upload must have landed
        self.op_ms = {k: 0.0 for k in ops}
        self.ibwt_us = [0, 0]
        self.phases = []
        self.sentinel = None

    def _timed(self, name, f):
        t0 = time.perf_counter()
        r = f()
        self.op_ms[name] += (time.perf_counter() - t0) * 1e3
        return r

    def step(self):
        c, n, ops = self.ctx, self.n, self.ops
        self._timed("sa", lambda: c.make_sa(self.d_text, n, self.d_sa))
        self.phases.append(c.timings())
        sent = None
        if "fbwt" in ops:      # the forward transform as ONE call (own suffix-array build + BWT, reference cpp:1771-1817)
            sent = self._timed("fbwt", lambda: c.forward_bwt(self.d_text, n, self.d_bwt))
        elif self.d_bwt is not None:
            sent = self._timed("bwt" if "bwt" in ops else "ibwt", lambda: c.bwt_from_sa(self.d_text, n, self.d_sa, self.d_bwt))
        if "ibwt" in ops:
            self._timed("ibwt", lambda: c.inverse_bwt(self.d_bwt, n, sent, self.d_inv))
            tmi = c.timings()
            self.ibwt_us[0] += tmi.ibwt_walk_us; self.ibwt_us[1] += tmi.ibwt_total_us
        if "lcp" in ops:
            self._timed("lcp", lambda: c.lcp(self.d_text, n, self.d_sa, self.d_lcp))
        self.sentinel = sent

    def run(self, steps, warmup):
        torch, dev = self.torch, self.dev
        for _ in range(warmup):
            self.step()
        self.phases.clear()
        for k in self.op_ms:
            self.op_ms[k] = 0.0
        self.ibwt_us = [0, 0]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize(dev)
        self.dt = time.perf_counter() - t0
        self.steps = steps
        return self.dt

    def validate(self):
        """On-device checker + the round trip; then the reference hashes when this exact input has a golden entry."""
        ok = self.ctx.validate_sa(self.d_text, self.n, self.d_sa) == 0
        if self.d_inv is not None:
            ok = ok and bool(self.torch.equal(self.d_inv, self.d_text[:self.n]))          # config 4: the round trip restores the text
        against = "on-device checker (adjacent-pair order + permutation)" + (" + round trip" if self.d_inv is not None else "")
        e = golden_entry(self.workload, self.seed, self.n)
        if e is not None:
            try:
                gok, checked = check_golden(e, self.d_sa, self.d_bwt, self.sentinel, self.d_lcp)
                ok = ok and gok
                against = "reference hash (FNV-1a-64 of " + ", ".join(checked) + "; tests/golden/golden_full.json) + " + against
            except Exception as ex:  # noqa: BLE001
                against += f" (reference hash not checked: {ex})"
        return bool(ok), against


def kernel_table(phases, K, n, workload, ops, ibwt_us):
    """Per-launch device time (HIP events on the engine's stream) and ALGORITHMIC bytes."""
    m = phases[-1].m
    avg = lambda f: sum(getattr(p, f) for p in phases) / K   # noqa: E731
    refilled = sum(p.gathered_records for p in phases) / K
    # two-stage build (text-like inputs): the sort phases below handle the B* suffixes only; the others are induced
    mstar = phases[-1].bstar_suffixes
    two_stage = mstar > 0
    ms_ = mstar if two_stage else m          # suffixes the sort phases handle
    # small alphabets: the key of the first gather round travels with the records through round 0: 4 bytes more per
    # suffix written by the scatter, 8 more moved by the partition level
    k1 = 1 if phases[-1].key1_records > 0 else 0
    kern = {
        "k_hist16": (avg("hist16_ms"), n),
        "k_scatter0": (avg("scatter0_ms"), n + (8 + 4 * k1) * ms_),
        "k_partition(level 1)": (avg("scatter1_ms"), (16 + 8 * k1) * ms_),
        ("bucket sort (LDS sorts of the two-byte buckets)" if workload == "random" else "round-0 LDS sorts (k_sort_mid/k_sort_tiny/bucket sort)"): (avg("bucket_sort_ms"), 12 * ms_),
    }
    if two_stage:
        rows_read = phases[-1].b_suffixes + n        # pass B reads the B rows, pass A every row
        # one text fetch per B* suffix and per induced suffix that has used its characters up: the rows carry three plain bytes, or - for
        # alphabets of up to 16 byte values (a DNA) - up to nine characters as dense numbers: every third / ninth row of a chain
        fetches = mstar + (n - mstar) / (9 if workload.startswith("dna") else 3)
        if os.environ.get("MSUFSORT_HIP_IND_CLASSIC"):
            # three kernels per level: rows read twice (count + scatter: 4 B index + 4 B characters each time)
            kern["induction (k_ind_count + k_ind_scan + k_ind_scatter)"] = (avg("other_ms"), int(16 * rows_read + 8 * (n - mstar) + 4 * fetches))
        else:
            # single-pass levels: every source row read once (4 B index + 4 B characters), every induced row written once (8 B)
            kern["induction (k_ind_fused + k_ind_small)"] = (avg("other_ms"), int(8 * rows_read + 8 * (n - mstar) + 4 * fetches))
    # (small alphabets: round 1 receives its records WITH their keys - no gather, but the same 16 bytes per record: record read 8,
    # row / next record written 8 - and its time is part of refine_ms: its records are counted here, not in gathered_records)
    keyed = sum(p.unresolved_after_round0 for p in phases) / K if k1 else 0
    if refilled + keyed > 0.01 * m:
        # per still-tied suffix and key round: index read (4) + key (8) + index written (4)
        kern["key rounds (k_refill + k_partition levels + LDS sorts)"] = (avg("refine_ms"), int(16 * (refilled + keyed)))
    if "ibwt" in ops and ibwt_us[0]:
        kern["k_ibwt_walk"] = (ibwt_us[0] / K / 1e3, 9 * n)       # n hops x 8 B entry + n bytes written
    return kern, avg, two_stage, mstar


def build_id():
    from msufsort_amd import _lib
    try:
        return _lib.lib().msufsort_hip_build_id().decode()
    except Exception:  # noqa: BLE001
        return None


def traffic_lookup(workload, n, kernel):
    """HBM bytes per launch of `kernel`: NOT measured in this run - copied from the committed PMC passes of the same command
    (profiles/pmc_traffic.json), and only when they were collected with THIS build of the library: a kernel change without a
    new PMC pass leaves no stale figure in the line (round-3 review)."""
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        e = pt.get("entries", {}).get(f"{workload}@{n}") or pt.get("entries", {}).get(workload)
        if not e or e.get("n") != n or kernel not in e["kernels"]:
            return None, None
        if pt.get("build_id") != build_id():
            return None, f"dropped: profiles/pmc_traffic.json was collected with library build {pt.get('build_id')}, this run loaded {build_id()}"
        return e["kernels"][kernel], (e.get("source", "profiles/pmc_traffic.json") + " (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE; not measured in "
                                      "this run; same library build " + str(pt.get("build_id")) + ")")
    except Exception:  # noqa: BLE001
        return None, None


def roofline_of(kern, n, workload, single_gpu):
    dom = max(kern, key=lambda k: kern[k][0])
    dms, dbytes = kern[dom]
    ach = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    traffic, traffic_source = traffic_lookup(workload, n, dom) if single_gpu else (None, None)
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "launch_ms": round(dms, 4), "algorithmic_bytes": int(dbytes)}


def kernels_json(kern):
    out = {k: {"ms": round(v[0], 4), "algorithmic_GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None,
               "frac_of_hbm_peak": round(v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[0] > 0 else None}
           for k, v in kern.items()}
    # a fraction above 1 means a kernel is billed for bytes it did not move in the time it is billed for (a phase timer that does
    # not cover it): such an entry is reported as an error with its fraction withheld - the validated headline is still printed
    for k, v in out.items():
        if v["frac_of_hbm_peak"] is not None and v["frac_of_hbm_peak"] > 1.0:
            v["error"] = f"algorithmic bandwidth {v['algorithmic_GBps']} GB/s is above the HBM peak: the phase timer does not cover this kernel"
            v["frac_of_hbm_peak"] = None
    return out


def random_access_ceiling(device_index):
    """What the memory system gives a kernel whose every access is another line, measured LIVE on this GPU (tools/micro/window_gather.hip
    built as build/librandom_access_probe.so by __graft_entry__.build(); 2^27 hashed indices per launch, best of 3): byte reads from a
    1 GiB and a 4 GiB window, 4-byte writes into 256 MiB and 1 GiB.  The gather-bound kernels of the text path (key rounds, induction,
    k_ibwt_walk, k_lcp) sit under THIS ceiling - ~52 G line requests/s whatever the window - not under the 8 TB/s of streaming reads:
    their fractions of the HBM peak are one 64-byte request per useful byte, their fractions of this figure say how well they hide
    the latency.  Measurement infrastructure: nothing in the product library calls it."""
    import ctypes as C
    path = os.path.join(ROOT, "build", "librandom_access_probe.so")
    try:
        f = C.CDLL(path).msufsort_probe_random_access
    except (OSError, AttributeError) as e:
        return {"error": f"{path}: {e} (run __graft_entry__.build())"}
    f.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = {"unit": "G accesses/s", "accesses_per_launch": 1 << 27, "source": "tools/micro/window_gather.hip, live in this run"}
    for name, window, mode in (("reads_1GiB_window", 1 << 30, 0), ("reads_4GiB_window", 4 << 30, 0), ("writes_256MiB_window", 256 << 20, 2), ("writes_1GiB_window", 1 << 30, 2)):
        v = C.c_double(0)
        out[name] = round(v.value, 1) if f(device_index, window, 1 << 27, mode, 3, C.byref(v)) == 0 else None
    return out


def request_bound(ceiling, key, accesses, ms, note):
    """One gather-bound phase against the measured random-access ceiling: accesses / time, as a fraction of ceiling[key]."""
    rate = accesses / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
    top = ceiling.get(key) if isinstance(ceiling, dict) else None
    return {"random_accesses": int(accesses), "ms": round(ms, 3), "G_per_s": round(rate, 1) if rate else None, "ceiling": key, "ceiling_G_per_s": top,
            "frac_of_measured_ceiling": round(rate / top, 3) if rate and top else None, "counted": note}


def config2_line(M, torch, ctx, dev, steps, no_cpu):
    """BASELINE config 2: suffix array of 256 MiB of uniform random bytes (splitmix64 seed 12345, n = 2^28) on one GPU, resident in
    HBM, timed like the headline (reference main.cpp:386,440-442: clock around the call); reference hash from golden_full.json."""
    n, seed = 1 << 28, 12345
    S = Single(M, torch, ctx, dev, "random", seed, n, ["sa"])
    dt = S.run(steps, 2)
    ok, against = S.validate()
    kern, avg, _, _ = kernel_table(S.phases, steps, n, "random", ["sa"], [0, 0])
    radix_ms = kern["k_hist16"][0] + kern["k_scatter0"][0]
    cfg2 = {"workload": f"random bytes (splitmix64 seed {seed}), n={n}, int32 SA, one GPU", "valid": ok, "valid_against": against, "steps": steps,
            "sa_ms": round(dt / steps * 1e3, 3), "sa_MBps": round(n / (dt / steps) / 1e6, 1), "sa_device_ms": round(avg("total_ms"), 3),
            "roofline": roofline_of(kern, n, "random", True), "kernels": kernels_json(kern),
            "radix_pass": {"read_bytes": 2 * n, "ms": round(radix_ms, 4),
                           "read_frac_of_hbm_peak": round((2 * n / (radix_ms * 1e-3) / 1e9) / HBM_PEAK_GBS, 4) if radix_ms > 0 else None},
            "end_to_end": {"compulsory_bytes": 5 * n + 4, "frac_of_hbm_peak": round(((5 * n + 4) / (dt / steps) / 1e9) / HBM_PEAK_GBS, 5)},
            "radix_bits": int(S.phases[-1].radix_bits), "bucket_sort_handed_back": int(S.phases[-1].bucket_sort_handed_back)}
    del S
    torch.cuda.empty_cache()
    if not no_cpu:
        try:
            cfg2["cpu_baseline"] = cpu_baseline(n, seed, "random", "sa", n, counts=[min(16, os.cpu_count() or 1)])
        except Exception as e:  # noqa: BLE001
            cfg2["cpu_baseline"] = {"error": str(e)}
    return cfg2, ok


def config5_line(M, torch, ctx, dev, no_cpu):
    """BASELINE config 5's workload at its full size on the ONE GPU of this run: 8 GiB of DNA with long tandem repeats (gen.dna_tandem_bytes
    seed 9, n = 2^33: the stream whose first 2^28 bytes are the golden vector), int64 rows (wide engine: 40-bit indices; the reference's
    int32 index with two flag bits stops at 2^30, msufsort.h:47, 84-93), the 8 GPUs' key-range shards as 32 logical shards that take
    turns on the device, distributed prefix doubling between them.  Timed: the SECOND build (every buffer in place; the first one
    allocates ~200 GB), clock around the call.  `valid`: the 64-bit on-device checker (order + permutation) on the full array, and
    the input's first 2^28 bytes hashed against the golden vector's input.  The 8-GPU form of this config is `bench.py --gpus 8
    --workload dna_tandem --size 8589934592` (dist.py)."""
    from msufsort_amd import gen
    n, seed, shards = 1 << 33, 9, 32
    ctx.trim()
    torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info(dev)
    if free_b < (250 << 30):
        return {"skipped": f"needs ~250 GiB of free HBM for the whole 8 GiB job on one GPU; {round(free_b / 2**30, 1)} GiB free of {round(total_b / 2**30, 1)}"}, True
    t0 = time.perf_counter()
    t = gen.dna_tandem_bytes(n, seed)
    g = golden_entry("dna_tandem", seed, 1 << 28)
    import oracle          # (the checker's hash routine, outside every timed part)
    prefix_ok = g is None or ("%016x" % oracle.fnv1a64(t[: 1 << 28])) == g["input_fnv"]
    d = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    for s0 in range(0, n, 1 << 30):
        d[s0:s0 + (1 << 30)] = torch.from_numpy(t[s0:s0 + (1 << 30)]).to(dev)
    del t
    torch.cuda.synchronize(dev)
    gen_s = time.perf_counter() - t0
    sa = torch.empty(n + 1, dtype=torch.int64, device=dev)
    t1 = time.perf_counter()
    ctx.make_sa_i64(d, n, sa, n_shards=shards)
    first_s = time.perf_counter() - t1
    sa.fill_(-1)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    ctx.make_sa_i64(d, n, sa, n_shards=shards)
    second_s = time.perf_counter() - t2
    tm = ctx.timings()
    ctx.trim()
    t3 = time.perf_counter()
    errs = ctx.validate_sa(d, n, sa, index_bytes=8)
    check_s = time.perf_counter() - t3
    ok = errs == 0 and int(sa[0]) == n and prefix_ok
    G = int(tm.logical_shards)
    # ALGORITHMIC bytes: the text is counted once (plan) and read by every shard's level-0 scatter; wide records are 8
    # bytes, rows 8; a key round moves 16 + 4 bytes per still-tied suffix (index 8 -> record 8, key 8, row 8: billed as SURVEY's 16 + the wider
    # index); a doubling step reads row + group head (12), gathers the rank of suffix + h (8) and writes row + head (12) per row it scans
    m = int(tm.m)
    dbl_rows = int(tm.doubling_records)
    # (the plan - the 16-bit histogram of the text and the deeper histograms of its heavy two-byte keys, ~0.13 s - runs once per build in
    # front of the shards and has no phase timer of its own: it is inside sa_ms / sa_device_ms, not in this table)
    kern = {f"k_scatter0 (x{G} shards)": (tm.scatter0_ms, G * n + 8 * m),
            "k_partition(level 1)": (tm.scatter1_ms, 16 * m),
            "round-0 LDS sorts (k_sort_mid/k_sort_tiny)": (tm.bucket_sort_ms, 16 * m),
            "key rounds (k_refill + k_partition levels + LDS sorts)": (tm.refine_ms, 24 * int(tm.gathered_records)),
            "distributed prefix doubling (k_import_groups, k_chain_resolve, k_refill_rows, LDS sorts, k_emit_updates, k_apply_updates)": (tm.other_ms, 32 * dbl_rows)}
    kern = {k: v for k, v in kern.items() if v[0] > 0}
    cfg5 = {"workload": f"dna_tandem (seed {seed}), n={n} (8 GiB), int64 SA, {G} logical key-range shards on one GPU (wide engine), distributed prefix doubling",
            "valid": bool(ok), "valid_against": "64-bit on-device checker (adjacent-pair order by ranks + permutation) on all 2^33 + 1 rows; the input's first 2^28 bytes "
                                                 "hash to the golden vector's input (tests/golden/golden_full.json: its rows are pinned by the reference in test_full_size_golden)",
            "checker_errors": int(errs), "sa_ms": round(second_s * 1e3, 1), "sa_MBps": round(n / second_s / 1e6, 1), "first_build_ms_with_allocations": round(first_s * 1e3, 1),
            "sa_device_ms": round(tm.total_ms, 1), "generate_and_upload_s": round(gen_s, 1), "check_s": round(check_s, 1),
            "logical_shards": G, "stop_depth": int(tm.stop_depth), "doubling_steps": int(tm.doubling_rounds), "doubling_ms": round(tm.other_ms, 1),
            "doubling_rows_scanned": dbl_rows, "key_rounds_gathered_records": int(tm.gathered_records), "unresolved_after_round0": int(tm.unresolved_after_round0),
            "phases_ms": {k: round(v[0], 2) for k, v in kern.items()}, "kernels": kernels_json(kern),
            "roofline": roofline_of(kern, n, "dna_tandem", True)}
    del sa, d
    ctx.trim()
    torch.cuda.empty_cache()
    if not no_cpu:
        try:      # the reference cannot index 8 GiB (msufsort.h:47): a bounded sample of the same stream - its first 2^28 bytes, the golden vector
            cfg5["cpu_baseline"] = cpu_baseline(1 << 28, seed, "dna_tandem", "sa", n, counts=[min(16, os.cpu_count() or 1)])
        except Exception as e:  # noqa: BLE001
            cfg5["cpu_baseline"] = {"error": str(e)}
    return cfg5, ok


def config_lines(M, torch, ctx, dev, steps, no_cpu):
    """BASELINE configs 3 and 4 under the same clock as the headline: text, 2^30 - 1 bytes (the reference's ceiling): SA,
    forward BWT as one call, inverse BWT, LCP - all resident in HBM; reference hashes from tests/golden/golden_full.json."""
    n, seed = (1 << 30) - 1, 3
    ops = ["sa", "fbwt", "ibwt", "lcp"]
    S = Single(M, torch, ctx, dev, "text", seed, n, ops)
    S.run(steps, 1)
    ok, against = S.validate()
    K = steps
    kern, avg, two_stage, mstar = kernel_table(S.phases, K, n, "text", ops, S.ibwt_us)
    kj = kernels_json(kern)
    sa_ms, fb_ms, ib_ms, lcp_ms = (S.op_ms[k] / K for k in ops)
    sa_kern = {k: v for k, v in kern.items() if k != "k_ibwt_walk"}
    cfg3 = {"workload": f"text (seed {seed}), n={n}: suffix array, then the forward BWT as one call", "valid": ok, "valid_against": against,
            "sa_ms": round(sa_ms, 3), "forward_bwt_ms": round(fb_ms, 3), "sa_MBps": round(n / sa_ms / 1e3, 1), "forward_bwt_MBps": round(n / fb_ms / 1e3, 1),
            "steps": K, "roofline": roofline_of(sa_kern, n, "text", True), "kernels": {k: kj[k] for k in sa_kern},
            "sa_device_ms": round(avg("total_ms"), 3)}
    if two_stage:
        cfg3["two_stage"] = {"bstar_suffixes": int(mstar), "induced_suffixes": int(n - mstar), "front_ms": round(avg("front_ms"), 3),
                             "induction_ms": round(avg("other_ms"), 3), "level_launches": int(S.phases[-1].induction_launches)}
    cfg3["fallbacks"] = int(sum(p.fallbacks & 1 for p in S.phases))
    cfg3["key1_records"] = int(S.phases[-1].key1_records)          # suffixes whose first gather round needed no gather
    cfg3["gathered_records"] = int(sum(p.gathered_records for p in S.phases) / K)
    walk_ms = S.ibwt_us[0] / K / 1e3
    cfg4 = {"workload": f"forward BWT of that text -> inverse BWT, n={n}, output compared with the text on the device", "valid": ok,
            "inverse_bwt_ms": round(ib_ms, 3), "inverse_bwt_MBps": round(n / ib_ms / 1e3, 1), "round_trip_ms": round(fb_ms + ib_ms, 3), "steps": K,
            "device_total_ms": round(S.ibwt_us[1] / K / 1e3, 3),
            "roofline": {"bound": "hbm", "kernel": "k_ibwt_walk", "achieved": round(9 * n / (walk_ms * 1e-3) / 1e9, 1) if walk_ms else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(9 * n / (walk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if walk_ms else None,
                         "traffic": traffic_lookup("text", n, "k_ibwt_walk")[0], "traffic_source": traffic_lookup("text", n, "k_ibwt_walk")[1],
                         "launch_ms": round(walk_ms, 3), "algorithmic_bytes": 9 * n,
                         "line_GBps": round(128 * n / (walk_ms * 1e-3) / 1e9, 1) if walk_ms else None},
            "lcp_ms": round(lcp_ms, 3)}
    # the gather-bound phases against the random-access ceiling measured in this run (their HBM fractions above are one 64-byte
    # request per few useful bytes by construction)
    ceil_ = random_access_ceiling(dev.index or 0)
    cfg3["random_access_ceiling"] = ceil_
    rb3 = {}
    if two_stage:
        rb3["induction"] = request_bound(ceil_, "reads_1GiB_window", mstar + (n - mstar) / 3, avg("other_ms"),
                                         "one 4-byte text fetch per B* suffix and per third induced suffix (the rows carry three characters); the rest of the phase is "
                                         "sequential rows, ranking and the look-back between tiles")
    gathered = sum(p.gathered_records for p in S.phases) / K
    if gathered:
        rb3["key rounds"] = request_bound(ceil_, "reads_1GiB_window", gathered, avg("refine_ms"),
                                          "one text line per gathered record; the time also holds the sorts of those records and the round whose keys came with the records")
    cfg3["request_bound"] = rb3
    cfg4["request_bound"] = {
        "k_ibwt_walk": request_bound(ceil_, "reads_4GiB_window", n, walk_ms, "one hop = one dependent 8-byte read from the 4 n-byte link table"),
        "k_lcp": request_bound(ceil_, "reads_1GiB_window", n, lcp_ms, "at least one text line per row (32 bytes from a random offset: 1.4 lines on average, PMC)")}
    if not no_cpu:
        try:
            cfg3["cpu_baseline"] = cpu_baseline(n, seed, "text", "fbwt", n)          # the WHOLE 2^30 - 1 text, like the GPU legs beside it (round-4 review)
        except Exception as e:  # noqa: BLE001
            cfg3["cpu_baseline"] = {"error": str(e)}
        try:       # the reference's inverse transform (cpp:1820-2103) on the same sample
            cfg4["cpu_baseline"] = cpu_baseline(n, seed, "text", "ibwt", n)
        except Exception as e:  # noqa: BLE001
            cfg4["cpu_baseline"] = {"error": str(e)}
    del S
    torch.cuda.empty_cache()
    return {"cfg3": cfg3, "cfg4": cfg4}, ok


def metric_label(workload, ops, n):
    """The headline string only for the headline input (BASELINE.json: SA build on 1 GiB random bytes = n 2^30 - 1, the oracle's
    ceiling); every other size / workload / op names itself (round-4 review: a 256 MiB run carried the 1 GiB label)."""
    if workload == "random" and ops == ["sa"] and n == (1 << 30) - 1:
        return "MB/s input for SA build on 1 GiB random bytes"
    return f"MB/s input for {'+'.join(ops)} on {workload} (n={n})"


def rccl_ranks_of(backend, world):
    """`config.rccl_ranks`: ranks of an RCCL communicator - None when the process group is not RCCL (the gloo test hook)."""
    return world if backend == "nccl" else None


def multi_gpu_budget(n, world, index_bytes, rows_max, two_stage, want_bwt, pipelined):
    """HBM bytes one rank of `bench.py --gpus N` allocates.  n = 2^33, N = 8, int64 rows: text 8 GiB + rows
    64 GiB + rank replica 64 GiB + sort workspace ~70 GiB + group heads / windows ~15 GiB = ~222 GiB of the 288 GB (268 GiB)."""
    from msufsort_amd import dist as mdist
    b = {"text": n + 64, "rows": (n + 1) * index_bytes * (2 if pipelined else 1), "group_heads": max(rows_max, 1) * 4,
         "sort_workspace": int((70 if index_bytes == 8 else 82) * (rows_max + rows_max // 8 + (2 << 20))) + (300 << 20),
         "doubling (rank replica + update / group-head windows; only for inputs with deep ties)": mdist.ShardState.bytes_needed(n, rows_max, world, index_bytes),
         "two_stage (sorted B* + induction workspace)": (n // 2 + 2) * 4 + 9 * n if two_stage else 0,
         "bwt": (n + rows_max) if want_bwt else 0}
    b["total"] = sum(b.values())
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=(1 << 30) - 1)   # 2^30-1: the oracle's ceiling
    ap.add_argument("--workload", default="random")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--op", default="sa")
    ap.add_argument("--cpu-sample", type=int, default=0, help="bytes of the stream the CPU baseline sorts (0: the whole input)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip the config 3 / 4 lines")
    ap.add_argument("--no-host", action="store_true", help="skip the host-pointer (PCIe inclusive) legs")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the config 5 line (8 GiB of tandem-repeat DNA on the one GPU: ~250 GiB of HBM, ~1 min)")
    ap.add_argument("--index", default="auto", choices=["auto", "int32", "int64"], help="N > 1: row width. auto = int64 (wide engine, 40-bit indices) beyond 2^31 - 2 bytes "
                    "(BASELINE config 5), int32 otherwise; int64 on a small input is the parity-test form of config 5")
    ap.add_argument("--check-reference", action="store_true", help="N > 1, after the timed region: rank 0 compares every row (and the BWT) with the CPU checker's (oracle/)")
    ap.add_argument("--check-single", action="store_true", help="N > 1, after the timed region: rank 0 rebuilds the array with the single-process entry point "
                    "(msufsort_hip_make_sa_i64_dev / _i32_dev: the path the parity tests pin to the reference) and compares every row (and the BWT) on the device - "
                    "the check that still works beyond the reference's 2^30 - 1 bytes")
    ap.add_argument("--two-stage", type=int, default=0, help="N > 1: 0 = text-like inputs take the sharded B* sort + induction on every rank "
                    "(the library's size / alphabet policy), 1 = whenever possible, -1 = never (sort-all shards)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}", file=sys.stderr)
        return 2

    if os.environ.get("MSUFSORT_BENCH_DEBUG_HANG"):          # every thread's stack on stderr after that many seconds (and again, repeatedly)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MSUFSORT_BENCH_DEBUG_HANG"]), repeat=True, file=sys.stderr)

    import torch

    import msufsort_amd as M
    from msufsort_amd import dist as mdist
    from msufsort_amd import gen

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ops = [x for x in args.op.split(",") if x]
    dist_path = world > 1 or bool(os.environ.get("MSUFSORT_BENCH_FORCE_DIST"))
    assert ops and ops[0] == "sa" and set(ops) <= {"sa", "bwt", "fbwt", "ibwt", "lcp"} and (not dist_path or ops in (["sa"], ["sa", "fbwt"])), "--op sa[,bwt][,fbwt][,ibwt][,lcp] (N > 1: sa or sa,fbwt)"
    n = args.size
    headline = args.workload == "random" and ops == ["sa"]
    metric = metric_label(args.workload, ops, n)

    # MSUFSORT_BENCH_FORCE_DIST=1 (test hook): a one-rank run takes the N > 1 code path - the one way a one-GPU box can put the
    # process-group calls of that path (init with device_id, broadcast, all-reduce, barrier) through RCCL itself
    if world == 1 and not os.environ.get("MSUFSORT_BENCH_FORCE_DIST"):
        dev = torch.device("cuda", local)
        ctx = M.DeviceContext(local, n)
        S = Single(M, torch, ctx, dev, args.workload, args.seed, n, ops)
        dt = S.run(args.steps, args.warmup)
        ok, against = S.validate()
        K = args.steps
        kern, avg, two_stage, mstar = kernel_table(S.phases, K, n, args.workload, ops, S.ibwt_us)
        radix_ms = kern["k_hist16"][0] + kern["k_scatter0"][0]
        out = {
            "metric": metric, "value": round(n / (dt / K) / 1e6, 2), "unit": "MB/s",
            "n_gpus": 1, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "valid": ok, "valid_against": against,
            "config": {"workload": f"{args.workload} bytes (splitmix64 seed {args.seed}), n={n}, int32 SA, one GPU", "n": n, "index": "int32", "ops": ops,
                       "allgatherv": None, "rccl_ranks": None, "pipelined": False},
            "roofline": roofline_of(kern, n, args.workload, True),
            "radix_pass": {"read_bytes": 2 * n, "ms": round(radix_ms, 4),
                           "read_frac_of_hbm_peak": round((2 * n / (radix_ms * 1e-3) / 1e9) / HBM_PEAK_GBS, 4) if radix_ms > 0 else None,
                           "note": "SURVEY 8(d) literal definition: 2n read bytes over k_hist16 + k_scatter0; the scatter is write bound (n + 8m bytes), see kernels"},
            "end_to_end": {"compulsory_bytes": 5 * n + 4, "frac_of_hbm_peak": round(((5 * n + 4) / (dt / K) / 1e9) / HBM_PEAK_GBS, 5)},
            "kernels": kernels_json(kern),
            "phases_ms": {k: round(v[0], 4) for k, v in kern.items()} | {"refine": round(avg("refine_ms"), 4), "device_total": round(avg("total_ms"), 4)},
            "fallbacks": int(sum(p.fallbacks & 1 for p in S.phases)),
            "key1_records": int(S.phases[-1].key1_records), "gathered_records": int(sum(p.gathered_records for p in S.phases) / K),
        }
        if two_stage:
            out["two_stage"] = {"bstar_suffixes": int(mstar), "induced_suffixes": int(n - mstar), "front_ms": round(avg("front_ms"), 3),
                                "induction_ms": round(avg("other_ms"), 3), "level_launches": int(S.phases[-1].induction_launches)}
        if len(ops) > 1:
            out["ops_ms"] = {k: round(v / K, 3) for k, v in S.op_ms.items()}
            if "ibwt" in ops:
                out["ibwt"] = {"walk_ms": round(S.ibwt_us[0] / K / 1e3, 3), "device_total_ms": round(S.ibwt_us[1] / K / 1e3, 3),
                               "walk_sector_GBps": round(64 * n / (S.ibwt_us[0] / K / 1e6) / 1e9, 1) if S.ibwt_us[0] else None}
        del S
        torch.cuda.empty_cache()
        if not args.no_host:
            # the drop-in's own numbers: host pointers in and out, result allocated inside the timed call (never `value`)
            try:
                floor = pcie_floor(torch, dev, n)
                out["end_to_end_host"] = end_to_end_host(torch, dev, args.workload, args.seed, n, floor, golden_entry(args.workload, args.seed, n))
            except Exception as e:  # noqa: BLE001
                out["end_to_end_host"] = {"error": repr(e)}
        if headline and n == (1 << 30) - 1 and not args.no_configs:
            try:
                out["configs"], cok = config_lines(M, torch, ctx, dev, 3, args.no_cpu)
                ok = ok and cok
            except Exception as e:  # noqa: BLE001
                out["configs"] = {"error": repr(e)}
            for name, fn in (("cfg2", lambda: config2_line(M, torch, ctx, dev, 5, args.no_cpu)), ("cfg5", lambda: config5_line(M, torch, ctx, dev, args.no_cpu))):
                if name == "cfg5" and args.no_cfg5:
                    continue
                try:
                    out["configs"][name], cok = fn()
                    ok = ok and cok
                except Exception as e:  # noqa: BLE001
                    out["configs"][name] = {"error": repr(e)}
                    ok = False
            out["configs"] = {k: out["configs"][k] for k in sorted(out["configs"], key=lambda k: (not k.startswith("cfg"), k))}
            if not args.no_host and "error" not in out["configs"]:
                try:
                    floor = out.get("end_to_end_host", {}).get("pcie_floor") or pcie_floor(torch, dev, (1 << 30) - 1)
                    out["configs"]["end_to_end_host"] = end_to_end_host(torch, dev, "text", 3, (1 << 30) - 1, floor, golden_entry("text", 3, (1 << 30) - 1))
                except Exception as e:  # noqa: BLE001
                    out["configs"]["end_to_end_host"] = {"error": repr(e)}
        if not args.no_cpu:
            try:
                sample = args.cpu_sample or n
                out["cpu_baseline"] = cpu_baseline(min(sample, n), args.seed, args.workload, "sa", n, all_threads=True, median_of=3 if headline else 1,
                                                    t1_prefix=(1 << 28) if headline else 0)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
        return 0 if ok else 1

    # ---------------------------------------------------------------- N > 1: one rank per GPU
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("MSUFSORT_BENCH_BACKEND", "nccl")     # "nccl" IS RCCL on ROCm
    if os.environ.get("MSUFSORT_BENCH_ONE_DEVICE"):                # test hook: all ranks share GPU 0
        local = 0
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    dev = torch.device("cuda", local)

    def refuse(msg, code=2):
        if rank == 0:
            print("bench.py: " + msg, file=sys.stderr)
        dist.destroy_process_group()
        return code

    # the communicator the exchange runs on must have exactly --gpus ranks: a line measured on fewer is refused
    backend_world = dist.get_world_size()
    if backend_world != args.gpus:
        return refuse(f"--gpus {args.gpus} but the {backend} process group has {backend_world} rank(s)")
    # row width: the reference's suffix_index is int32 with two flag bits (msufsort.h:47, 84-93: n < 2^30); int32 rows here hold
    # n <= 2^31 - 2, beyond that (BASELINE config 5: 8 GiB) the wide engine writes int64 rows
    int32_max_n = (1 << 31) - 2
    if args.index == "int32" and n > int32_max_n:
        return refuse(f"--index int32 holds n <= {int32_max_n}; n = {n} needs --index int64 (or auto)")
    wide = args.index == "int64" or (args.index == "auto" and n > int32_max_n)
    index_bytes, row_dt, index_name = (8, torch.int64, "int64") if wide else (4, torch.int32, "int32")
    want_bwt = "fbwt" in ops

    two_stage_possible = (not wide) and args.two_stage >= 0
    pipelined_ok = (not wide) and (not want_bwt)

    # memory: say what is needed and refuse BEFORE anything large is allocated or generated (n = 2^33 over 8 ranks: ~222 GiB per
    # GPU) - first with balanced shards assumed, again once the real slice bounds are known
    share = world if os.environ.get("MSUFSORT_BENCH_ONE_DEVICE") else 1          # (test hook: every rank allocates on the one GPU)

    def hbm_short(rows_max, allocated):
        budget = multi_gpu_budget(n, world, index_bytes, rows_max, two_stage_possible, want_bwt, pipelined_ok)
        free_b, total_b = torch.cuda.mem_get_info(dev)
        short = torch.tensor([1 if (budget["total"] - allocated) * share > free_b else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(short, op=dist.ReduceOp.MAX)
        if not int(short.item()):
            return budget, None
        gib = lambda v: round(v / 2**30, 2)  # noqa: E731
        return budget, (f"n = {n} with {index_name} rows over {world} rank(s) needs ~{gib(budget['total'])} GiB of HBM per GPU "
                        f"({', '.join(f'{k}: {gib(v)}' for k, v in budget.items() if k != 'total' and v)}); rank {rank} has {gib(free_b)} GiB free "
                        f"of {gib(total_b)}{' shared by ' + str(world) + ' ranks' if share > 1 else ''} - use more GPUs or a smaller --size")

    budget, msg = hbm_short((n + 1) // world + (n >> 6) + 1, 0)
    if msg:
        return refuse(msg)

    # the text is generated ONCE (rank 0) and replicated over the links ("H2D to one + broadcast")
    d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    bcast = not os.environ.get("MSUFSORT_BENCH_NO_BROADCAST")
    if rank == 0 or not bcast:
        t = gen.GENERATORS[args.workload](n, args.seed)
        for s0 in range(0, n, 1 << 30):
            d_text[s0:min(n, s0 + (1 << 30))] = torch.from_numpy(t[s0:min(n, s0 + (1 << 30))]).to(dev)
        del t
    torch.cuda.synchronize(dev)
    if bcast:
        dist.broadcast(d_text, src=0)
    torch.cuda.synchronize(dev)          # the engine works on its own HIP stream: the text must have landed
    ctx = M.DeviceContext(local, 0)
    # every rank sorts its key range as k sub-shards: sub-slice j travels (one group of sends / receives) while sub-shard j + 1 is sorted
    # (dist.build_sa_sharded, sub_bounds; reference: threads leave their partitions while others still sort, msufsort.cpp:1652-1683)
    k_sub = mdist.sub_shards_for(world, n)
    bounds, sub_bounds = mdist.plan_sub_bounds(ctx, d_text, n, world, k_sub)
    if k_sub <= 1:
        sub_bounds = None
    rows_max = max(bounds[g + 1] - bounds[g] for g in range(world))
    budget, msg = hbm_short(rows_max, n + 64)
    if msg:
        return refuse(msg)

    exchange = mdist.select_exchange(dist, dev)
    d_sa = torch.empty(n + 1, dtype=row_dt, device=dev)
    d_grp = torch.empty(max(rows_max, 1), dtype=torch.int32, device=dev)          # tie-group heads of MY slice only
    sa_bufs = [d_sa] + ([torch.empty(n + 1, dtype=row_dt, device=dev)] if pipelined_ok else [])
    d_bwt = torch.empty(n, dtype=torch.uint8, device=dev) if want_bwt else None
    d_row_bytes = torch.empty(max(rows_max, 1), dtype=torch.uint8, device=dev) if want_bwt else None
    pending = {"works": [], "buf": None, "last": d_sa, "k": 0, "sentinel": None}
    shard_state = mdist.ShardState()
    phases = []
    op_ms = {k: 0.0 for k in ops}

    # A step = ONE complete build: every rank sorts its key range, then the all-gatherv of the slices, finished on every
    # rank before the next step starts - nothing of one build overlaps another.  ms_per_step is therefore the LATENCY of a
    # build, and `value` the throughput a caller sees who needs each array before asking for the next (round-3 review:
    # a strong-scaling curve of a pipelined rate would flatter).
    d_bstar = torch.empty(n // 2 + 2, dtype=torch.int32, device=dev) if two_stage_possible else None
    ts_stats, bwt_stats, hist_stats = {}, {}, {}
    # the histogram a sharded build starts with is counted 1/N per rank (dist.plan_sharded): its all-reduce and all-gather get a
    # communicator of their own, so that in the pipelined figure they do not queue behind the previous build's slices
    hist_group = dist.new_group(ranks=list(range(backend_world))) if (mdist.sharded_hist_enabled(world, n) or k_sub > 1) else None

    def build_rows(out, gather_rows=True):
        """True: the two-stage sharded build ran (every rank holds ALL rows); False: the sort-all shards ran."""
        # text-like inputs: B* suffixes sorted by key-range shards, their slices exchanged, the rest induced on every rank
        # (declines - on every rank alike - for anything else: random bytes take the sort-all shards below)
        if d_bstar is not None and ts_stats.get("two_stage_status") != 1 and mdist.build_sa_two_stage_sharded(ctx, d_text, n, out, d_bstar, rank, world, dist, two_stage=args.two_stage, stats=ts_stats):
            return True
        mdist.build_sa_sharded(ctx, d_text, n, out, rank, world, dist, bounds, d_grp=d_grp, overlap=False, index_bytes=index_bytes, state=shard_state,
                               gather_rows=gather_rows, stats=hist_stats, hist_group=hist_group, sub_bounds=sub_bounds)
        return False

    def step():
        t0 = time.perf_counter()
        build_rows(sa_bufs[0])
        phases.append(ctx.timings())
        pending["last"] = sa_bufs[0]
        t1 = time.perf_counter()
        op_ms["sa"] += (t1 - t0) * 1e3
        if want_bwt:
            # the forward transform as ONE call (its own sort, like msufsort::forward_burrows_wheeler_transform, cpp:1771-1817): the
            # rows stay distributed, every rank gathers the bytes of its slice, n/G-byte slices travel instead of the rows
            if build_rows(sa_bufs[0], gather_rows=False):
                pending["sentinel"] = ctx.bwt_from_sa(d_text, n, sa_bufs[0], d_bwt, index_bytes)          # all rows are here already: no exchange
            else:
                pending["sentinel"] = mdist.forward_bwt_sharded(ctx, d_text, n, sa_bufs[0], bounds, rank, world, dist, d_bwt, d_row_bytes, index_bytes, stats=bwt_stats)
            op_ms["fbwt"] += (time.perf_counter() - t1) * 1e3

    # the pipelined flavour (secondary figure): two output buffers, the all-gatherv of build k travels while build k+1 is sorted
    def step_pipelined():
        out = sa_bufs[pending["k"] & 1]
        pending["k"] += 1
        works = mdist.build_sa_sharded(ctx, d_text, n, out, rank, world, dist, bounds, d_grp=d_grp, overlap=True, index_bytes=index_bytes, state=shard_state,
                                       stats=hist_stats, hist_group=hist_group, sub_bounds=sub_bounds)
        mdist.wait_all(pending["works"], pending["buf"])        # the previous exchange overlapped with this build
        pending["works"], pending["buf"], pending["last"] = works, out, out

    def drain():
        mdist.wait_all(pending["works"], pending["buf"])
        pending["works"] = []

    def barrier():
        dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    phases.clear()
    for k in op_ms:
        op_ms[k] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    x = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    dt = float(x.item())

    used_two_stage = ts_stats.get("two_stage_status") == 0
    # secondary, outside the timed region: (a) the pipelined rate, (b) where one build's latency goes (sort, then exchange) -
    # for the sort-all shards (a two-stage sharded build has no slice exchange to overlap: its secondary figures are the B*
    # exchange and this rank's device time)
    kp = max(2, min(args.steps, 5))
    if used_two_stage:
        # (the build's device clock runs across the exchange callback: what is left is this rank's own work)
        x = torch.tensor([dt / args.steps * 1e3, ts_stats.get("bstar_exchange_ms", 0.0), max(ctx.timings().total_ms - ts_stats.get("bstar_exchange_ms", 0.0), 0.0)],
                         dtype=torch.float64, device=dev)
        per = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(per, x)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        latency = {"latency_ms": round(float(x[0]), 3), "exchange_ms": round(float(x[1]), 3), "sort_ms": round(float(x[2]), 3), "pipelined": None,
                   "per_rank": {"latency_ms": [round(float(q[0]), 3) for q in per], "exchange_ms": [round(float(q[1]), 3) for q in per],
                                "sort_ms": [round(float(q[2]), 3) for q in per], "rows": [int(bounds[g + 1] - bounds[g]) for g in range(world)]}}
    if not used_two_stage:
        pipelined = None
        if pipelined_ok:
            step_pipelined(); drain(); barrier()
            tp0 = time.perf_counter()
            for _ in range(kp):
                step_pipelined()
            drain()
            barrier()
            xp = torch.tensor([time.perf_counter() - tp0], dtype=torch.float64, device=dev)
            dist.all_reduce(xp, op=dist.ReduceOp.MAX)
            pipelined_ms = float(xp.item()) / kp * 1e3
            pipelined = {"ms_per_build": round(pipelined_ms, 3), "MBps": round(n / pipelined_ms / 1e3, 1), "builds": kp,
                         "note": "build k+1 sorted while the all-gatherv of build k travels (two output buffers); NOT `value`"}
        lat, exc = [], []
        for _ in range(2):
            barrier()
            a = time.perf_counter()
            w = mdist.build_sa_sharded(ctx, d_text, n, sa_bufs[0], rank, world, dist, bounds, d_grp=d_grp, overlap=True, index_bytes=index_bytes, state=shard_state,
                                       stats=hist_stats, hist_group=hist_group, sub_bounds=sub_bounds)
            torch.cuda.synchronize(dev)
            b = time.perf_counter()
            mdist.wait_all(w, sa_bufs[0])
            barrier()
            c_ = time.perf_counter()
            lat.append((c_ - a) * 1e3); exc.append((c_ - b) * 1e3)
        x = torch.tensor([min(lat), min(exc), ctx.timings().total_ms], dtype=torch.float64, device=dev)
        per = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(per, x)                       # per-rank figures: is one shard slower than the others?
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        pending["last"] = sa_bufs[0]
        latency = {"latency_ms": round(float(x[0]), 3), "exchange_ms": round(float(x[1]), 3), "sort_ms": round(float(x[2]), 3),
                   "pipelined": pipelined,
                   # what the sorts alone sustain (every rank keeps its slice, nothing is gathered): NOT `value` - every GPU must take in
                   # (N-1)/N of the row array over xGMI for the all-gatherv the metric asks for, and that, not the sort, bounds the step
                   "sorts_only_MBps": round(n / (float(x[2]) * 1e-3) / 1e6, 1) if float(x[2]) > 0 else None,
                   "per_rank": {"latency_ms": [round(float(q[0]), 3) for q in per], "exchange_ms": [round(float(q[1]), 3) for q in per],
                                "sort_ms": [round(float(q[2]), 3) for q in per],
                                "rows": [int(bounds[g + 1] - bounds[g]) for g in range(world)]}}
    ok = True
    if rank == 0:
        ok = ctx.validate_sa(d_text, n, pending["last"], index_bytes) == 0     # on-device checker on the assembled array
        against = "on-device checker (adjacent-pair order + permutation) on the assembled array"
        if want_bwt:
            # the gathered byte slices against the transform read off the assembled rows on this device
            chk = torch.empty(n, dtype=torch.uint8, device=dev)
            s_chk = ctx.bwt_from_sa(d_text, n, pending["last"], chk, index_bytes)
            ok = ok and s_chk == pending["sentinel"] and bool(torch.equal(chk, d_bwt))
            against += " + BWT bytes and sentinel row equal to the transform read off the assembled rows"
            del chk
        if args.check_single:
            # the multi-process flow against the single-process build of the same library (pinned to the reference by tests/test_gpu_parity.py
            # and test_gpu_full.py; beyond 2^31 - 2 bytes by the 64-bit checker in tests/test_gpu_big.py): every row, on the device
            for k_ in ("isa", "grp_prev", "upd_local", "upd_all", "grp_all"):
                setattr(shard_state, k_, None)
            del sa_bufs[1:]
            pending["buf"] = None
            ctx.trim()
            torch.cuda.empty_cache()
            ref_rows = torch.empty(n + 1, dtype=row_dt, device=dev)
            if wide:
                ctx.make_sa_i64(d_text, n, ref_rows)
            else:
                ctx.make_sa(d_text, n, ref_rows)
            same = bool(torch.equal(ref_rows, pending["last"]))
            ok = ok and same
            against += " + every row equal to the single-process build's (" + ("msufsort_hip_make_sa_i64_dev" if wide else "msufsort_hip_make_sa_i32_dev") + ")"
            if want_bwt:
                ctx.trim()
                chk = torch.empty(n, dtype=torch.uint8, device=dev)
                s_chk = ctx.bwt_from_sa(d_text, n, ref_rows, chk, index_bytes)
                ok = ok and s_chk == pending["sentinel"] and bool(torch.equal(chk, d_bwt))
                against += " + BWT bytes and sentinel row equal to the transform of the single-process rows"
                del chk
            del ref_rows
            ctx.trim()
            torch.cuda.empty_cache()
        e = golden_entry(args.workload, args.seed, n)
        if e is not None and not wide:
            try:
                gok, checked = check_golden(e, pending["last"], d_bwt, pending["sentinel"])
                ok = ok and gok
                against = "reference hash (FNV-1a-64 of " + ", ".join(checked) + "; tests/golden/golden_full.json) + " + against
            except Exception as ex:  # noqa: BLE001
                against += f" (reference hash not checked: {ex})"
        if args.check_reference:
            # every row (and the BWT) against the CPU checker - the unmodified reference when oracle/_ref is here, else the C restatement
            import numpy as np

            import oracle
            th = gen.GENERATORS[args.workload](n, args.seed)
            use_ref = oracle.have_reference() and n < (1 << 30)
            want = oracle.ref_make_suffix_array(th, 4) if use_ref else oracle.make_suffix_array(th)
            ok = ok and bool((pending["last"].cpu().numpy().astype(np.int64) == np.asarray(want, dtype=np.int64)).all())
            if want_bwt:
                wb, ws = oracle.ref_forward_bwt(th, 4) if use_ref else oracle.forward_bwt(th)
                ok = ok and int(ws) == int(pending["sentinel"]) and bool((d_bwt.cpu().numpy() == wb).all())
            against = ("rows" + (" + BWT" if want_bwt else "") + " equal to " + ("the unmodified reference's (oracle/_ref)" if use_ref else "the C restatement's (oracle/)") + " + " + against)
        K = args.steps
        kern, avg, two_stage, mstar = kernel_table(phases, K, n, args.workload, ["sa"], [0, 0])
        # a rank reads the whole text (hist + scatter) but sorts only its shard: bill the record passes with the shard's suffixes
        if used_two_stage:
            # (the sort phases saw this rank's share of the B* suffixes; the induction ran over all rows on every rank)
            my = int(mstar // world)
            kern = {k: ((v[0], n + 8 * my) if k == "k_scatter0" else (v[0], 16 * my) if k.startswith("k_partition") else (v[0], 12 * my) if "LDS sorts" in k or "bucket sort" in k else v)
                    for k, v in kern.items() if not k.startswith("key rounds")}
        else:
            my = int(bounds[1] - bounds[0])
            kern = {"k_hist16": kern["k_hist16"], "k_scatter0": (kern["k_scatter0"][0], n + 8 * my),
                    "k_partition(level 1)": (kern["k_partition(level 1)"][0], 16 * my),
                    "bucket sort (rank 0's shard)": (avg("bucket_sort_ms"), (8 + index_bytes) * my)}
        # (key rounds / distributed doubling of a sharded build are several calls with their own timings: see "doubling")
        out = {
            "metric": metric, "value": round(n / (dt / K) / 1e6, 2), "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "valid": bool(ok), "valid_against": against,
            "config": {"workload": f"{args.workload} bytes (splitmix64 seed {args.seed}), n={n}, {index_name} SA, 4-byte-prefix range sharding x{world}",
                       "n": n, "index": index_name, "ops": ops, "allgatherv": exchange, "rccl_ranks": rccl_ranks_of(backend, backend_world), "ranks": backend_world,
                       "pipelined": False, "backend": backend,
                       "overlap": {"sub_shards_per_rank": k_sub,
                                   "what": ("every rank sorts its key range as %d sub-shards; sub-slice j is posted (grouped sends / receives) as soon as it is sorted and "
                                            "travels while sub-shard j + 1 is sorted; `value` stays the latency of complete builds" % k_sub) if k_sub > 1 else
                                           "none: the slice is posted when the whole key range is sorted (small input, one rank, or rows kept distributed)"},
                       "step": ("B* suffixes of my key range sorted + all-gatherv of the sorted-B* slices (4|B*| bytes) + induction of all rows on every rank" if used_two_stage else
                                "sort of my key range + all-gatherv of the slices, complete on every rank (latency of one build)") +
                               ("; then the forward BWT as one call: its own sharded sort, rows kept distributed, n/G-byte slices all-gathered" if want_bwt else "")},
            "roofline": roofline_of(kern, n, args.workload, False),
            "kernels": kernels_json(kern),
            "phases_ms": {k: round(v[0], 4) for k, v in kern.items()} | {"refine": round(avg("refine_ms"), 4), "device_total": round(avg("total_ms"), 4)},
            "allgatherv_bytes_per_rank": int((4 * mstar if used_two_stage else index_bytes * (n + 1)) * (world - 1) / world),
            "hbm_budget_GiB_per_rank": {k: round(v / 2**30, 3) for k, v in budget.items() if v},
        }
        out.update(latency)           # where the latency goes (sort / exchange, per rank) and the pipelined rate as a secondary figure
        if want_bwt:
            out["ops_ms"] = {k: round(v / K, 3) for k, v in op_ms.items()}
            out["forward_bwt"] = dict(bwt_stats, sentinel_row=pending["sentinel"], MBps=round(n / (op_ms["fbwt"] / K) / 1e3, 1) if op_ms["fbwt"] else None,
                                      exchanged="nothing beyond the sorted-B* slices (every rank induced all rows)" if used_two_stage else
                                      f"{int(n * (world - 1) / world)} bytes per rank (n/G-byte slices) instead of {int(index_bytes * (n + 1) * (world - 1) / world)} (rows)")
        out["histogram"] = (f"counted 1/{world} per rank, one all-reduce + one all-gather per build ({hist_stats['sharded_hist']} builds)" if hist_stats.get("sharded_hist")
                            else "replicated: every rank counts the whole text (below the size where counting 1/N pays - MSUFSORT_DIST_SHARDED_HIST=1 forces it -, a shard boundary inside a heavy two-byte key, or the two-stage build)")
        if shard_state.stats:
            out["doubling"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in shard_state.stats.items()}
        if ts_stats:
            out["two_stage_sharded"] = dict(ts_stats, note="status 0: B* suffixes sorted by key-range shards, 4|B*| bytes all-gathered, the rest induced on every rank; "
                                                         "1: declined on every rank (sort-all shards ran)")
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
