#!/usr/bin/env python3
"""bench.py - MB/s of input for the suffix-array build on uniform-random bytes (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--size BYTES] [--workload random|text|dna|dna_tandem] [--op sa[,bwt][,ibwt][,lcp]]

Default (what the driver runs): the headline, SA of 2^30 - 1 uniform random bytes.  `--workload text --op sa,bwt,ibwt`
is BASELINE config 3 + 4 (a step = SA build, BWT from the SA, inverse BWT; single GPU).

A "step" is one complete suffix-array build (16-bit radix histogram, two 8-bit scatter levels, LDS bucket
sorts, refinement rounds) of one synthetic input that is already resident in HBM.  N = 1: the whole
array on one MI355X.  N > 1 (launched by torch.distributed.run, one rank per GPU): the 16-bit key space
is split into N count-balanced ranges, every rank sorts its range into its slice of the full array and
the slices are exchanged with one all-gatherv (one group of direct sends/receives over RCCL/xGMI); total work
is fixed, so "scaling" is "strong".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(sample_bytes, seed, workload="random"):
    """The unmodified reference (oracle/_ref, "reference") or our C restatement ("port") timed on the
    host cores on a bounded sample of the same stream.  Checker/baseline only - never the product."""
    import numpy as np  # noqa: F401

    import oracle
    from msufsort_amd import gen
    t = gen.GENERATORS[workload](sample_bytes, seed)
    ncpu = os.cpu_count() or 1
    if oracle.have_reference():
        cores = max(1, min(32, ncpu))
        kind = "reference"
        runs = []
        for _ in range(3):                      # spin-wait pool: median of 3 runs on the same input
            t0 = time.perf_counter()
            oracle.ref_make_suffix_array(t, cores)
            runs.append(time.perf_counter() - t0)
        med = sorted(runs)[1]
        note = "median of 3 (runs: " + ", ".join(f"{sample_bytes / r / 1e6:.1f}" for r in runs) + " MB/s)"
    else:
        cores, kind = 1, "port"
        t0 = time.perf_counter()
        oracle.make_suffix_array(t)
        med = time.perf_counter() - t0
        note = "one run"
    return {"value": round(sample_bytes / med / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": kind,
            "sample": f"first {sample_bytes} bytes of the same stream, make_suffix_array wall time incl. SA allocation, {note}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=(1 << 30) - 1)   # 2^30-1: the oracle's ceiling
    ap.add_argument("--workload", default="random")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--op", default="sa")
    ap.add_argument("--cpu-sample", type=int, default=1 << 28)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    import msufsort_amd as M
    from msufsort_amd import dist as mdist
    from msufsort_amd import gen

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
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
    n = args.size
    t = gen.GENERATORS[args.workload](n, args.seed)
    d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_text[:n] = torch.from_numpy(t).to(dev)
    d_sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)          # the engine works on its own HIP stream: the upload must have landed
    ctx = M.DeviceContext(local, n if world == 1 else n // world + n // (8 * world) + (1 << 20))     # workspace: my shard's suffixes

    bounds = ctx.shard_bounds(d_text, n, world) if world > 1 else None
    exchange = mdist.select_exchange(dist, dev) if world > 1 else None
    d_grp = torch.empty(n + 1, dtype=torch.int32, device=dev) if world > 1 else None
    # N > 1: two output buffers, so the all-gatherv of build k can travel while build k+1 is being sorted
    # (every build and every exchange is complete before the closing barrier of the timed region)
    sa_bufs = [d_sa, torch.empty(n + 1, dtype=torch.int32, device=dev)] if world > 1 else [d_sa]
    pending = {"works": [], "buf": None, "last": d_sa, "k": 0}
    shard_state = mdist.ShardState() if world > 1 else None

    ops = [x for x in args.op.split(",") if x]
    assert ops and ops[0] == "sa" and set(ops) <= {"sa", "bwt", "fbwt", "ibwt", "lcp"} and (world == 1 or ops == ["sa"]), "--op sa[,bwt][,fbwt][,ibwt][,lcp] (N > 1: sa only)"
    d_bwt = torch.empty(n, dtype=torch.uint8, device=dev) if ("bwt" in ops or "ibwt" in ops or "fbwt" in ops) else None
    d_inv = torch.empty(n, dtype=torch.uint8, device=dev) if "ibwt" in ops else None
    d_lcp = torch.empty(n, dtype=torch.int32, device=dev) if "lcp" in ops else None
    op_ms = {k: 0.0 for k in ops}
    ibwt_us = [0, 0]

    def timed(name, f):
        t0 = time.perf_counter()
        r = f()
        op_ms[name] += (time.perf_counter() - t0) * 1e3
        return r

    def step():
        if world == 1:
            timed("sa", lambda: ctx.make_sa(d_text, n, d_sa))
            phases.append(ctx.timings())
            sent = None
            if "fbwt" in ops:      # the forward transform as ONE call (own suffix-array build + BWT, reference cpp:1771-1817)
                sent = timed("fbwt", lambda: ctx.forward_bwt(d_text, n, d_bwt))
            elif d_bwt is not None:
                sent = timed("bwt" if "bwt" in ops else "ibwt", lambda: ctx.bwt_from_sa(d_text, n, d_sa, d_bwt))
            if "ibwt" in ops:
                timed("ibwt", lambda: ctx.inverse_bwt(d_bwt, n, sent, d_inv))
                tmi = ctx.timings()
                ibwt_us[0] += tmi.reserved[3]; ibwt_us[1] += tmi.reserved[4]
            if "lcp" in ops:
                timed("lcp", lambda: ctx.lcp(d_text, n, d_sa, d_lcp))
            return
        out = sa_bufs[pending["k"] & 1]
        pending["k"] += 1
        works = mdist.build_sa_sharded(ctx, d_text, n, out, rank, world, dist, bounds, d_grp_full=d_grp, overlap=True, state=shard_state)
        mdist.wait_all(pending["works"], pending["buf"])        # the previous exchange overlapped with this build
        pending["works"], pending["buf"], pending["last"] = works, out, out

    def drain():
        mdist.wait_all(pending["works"], pending["buf"])
        pending["works"] = []

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    phases = []
    for _ in range(args.warmup):
        step()
    drain()
    phases.clear()
    for k in op_ms:
        op_ms[k] = 0.0
    ibwt_us[0] = ibwt_us[1] = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if world > 1:
            phases.append(ctx.timings())
    drain()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        x = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        dt = float(x.item())

    # N > 1: the timed loop pipelines build k+1 over the exchange of build k (throughput).  The latency of ONE build -
    # sort, then its exchange, nothing overlapped - is measured separately, outside the timed region.
    latency = None
    if world > 1:
        lat, exc = [], []
        for _ in range(2):
            barrier()
            a = time.perf_counter()
            w = mdist.build_sa_sharded(ctx, d_text, n, sa_bufs[0], rank, world, dist, bounds, d_grp_full=d_grp, overlap=True, state=shard_state)
            torch.cuda.synchronize(dev)
            b = time.perf_counter()
            mdist.wait_all(w, sa_bufs[0])
            barrier()
            c_ = time.perf_counter()
            lat.append((c_ - a) * 1e3); exc.append((c_ - b) * 1e3)
        x = torch.tensor([min(lat), min(exc)], dtype=torch.float64, device=dev)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        latency = {"latency_ms": round(float(x[0]), 3), "exchange_ms": round(float(x[1]), 3)}
        pending["last"] = sa_bufs[0]

    ok = True
    if rank == 0:
        ok = ctx.validate_sa(d_text, n, pending["last"]) == 0     # on-device checker on the assembled array
        if d_inv is not None:
            ok = ok and bool(torch.equal(d_inv, d_text[:n]))          # config 4: the round trip restores the text

    if rank == 0:
        K = args.steps
        ms_per_step = dt / K * 1e3
        m = phases[-1].m
        avg = lambda f: sum(getattr(p, f) for p in phases) / K   # noqa: E731
        # per-launch device time (HIP events on the engine's stream) and ALGORITHMIC bytes
        refilled = sum(p.reserved[2] for p in phases) / K
        # two-stage build (text-like inputs): the sort phases below handle the B* suffixes only; the others are induced
        mstar = phases[-1].reserved[5]
        two_stage = mstar > 0
        ms_ = mstar if two_stage else m          # suffixes the sort phases handle
        kern = {
            "k_hist16": (avg("hist16_ms"), n),
            "k_scatter0": (avg("scatter0_ms"), n + 8 * ms_),
            "k_partition(level 1)": (avg("scatter1_ms"), 16 * ms_),
            ("k_sort_fast2(bucket sort)" if args.workload == "random" else "round-0 LDS sorts (k_sort_mid/k_sort_tiny/k_sort_fast2)"): (avg("bucket_sort_ms"), 12 * ms_),
        }
        if two_stage:
            # rows read twice (count + scatter: 4 B index + 4 B characters), every induced row written once (8 B),
            # one 4-byte text fetch per B* suffix and per third induced suffix
            rows_read = phases[-1].reserved[7] + n
            kern["induction (k_ind_count + k_ind_scan + k_ind_scatter)"] = (avg("other_ms"), int(16 * rows_read + 8 * (n - mstar) + 4 * (mstar + (n - mstar) / 3)))
        if refilled > 0.01 * m:
            # per still-tied suffix and key round: index read (4) + key (8) + index written (4)
            kern["key rounds (k_refill + k_partition levels + LDS sorts)"] = (avg("refine_ms"), int(16 * refilled))
        if "ibwt" in ops and ibwt_us[0]:
            # n hops x 8 B entry + n bytes written
            kern["k_ibwt_walk"] = (ibwt_us[0] / K / 1e3, 9 * n)
        dom = max(kern, key=lambda k: kern[k][0])
        dms, dbytes = kern[dom]
        ach = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
        traffic, traffic_source = None, None
        try:   # HBM bytes per launch: NOT measured in this run - copied from the committed PMC passes of the same command
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("n") == n and world == 1 and args.workload == "random":
                traffic = pt["kernels"].get(dom)
                traffic_source = "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE; not measured in this run)"
        except Exception:  # noqa: BLE001
            traffic = None
        radix_ms = kern["k_hist16"][0] + kern["k_scatter0"][0]
        out = {
            "metric": "MB/s input for SA build on 1 GiB random bytes" if (args.workload == "random" and ops == ["sa"]) else f"MB/s input for {'+'.join(ops)} on {args.workload}",
            "value": round(n / (dt / K) / 1e6, 2),
            "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "valid": bool(ok),
            "config": {"workload": f"{args.workload} bytes (splitmix64 seed {args.seed}), n={n}, int32 SA, 16-bit-key range sharding x{world}",
                       "n": n, "index": "int32", "ops": ops, "allgatherv": exchange, "rccl_ranks": world if world > 1 else None,
                       "pipelined": world > 1},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "launch_ms": round(dms, 4), "algorithmic_bytes": int(dbytes)},
            "radix_pass": {"read_bytes": 2 * n, "ms": round(radix_ms, 4),
                           "read_frac_of_hbm_peak": round((2 * n / (radix_ms * 1e-3) / 1e9) / HBM_PEAK_GBS, 4) if radix_ms > 0 else None},
            "end_to_end": {"compulsory_bytes": 5 * n + 4, "frac_of_hbm_peak": round(((5 * n + 4) / (dt / K) / 1e9) / HBM_PEAK_GBS, 5)},
            "kernels": {k: {"ms": round(v[0], 4), "algorithmic_GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None,
                            "frac_of_hbm_peak": round(v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[0] > 0 else None}
                        for k, v in kern.items()},
            "phases_ms": {k: round(v[0], 4) for k, v in kern.items()} | {"refine": round(avg("refine_ms"), 4), "device_total": round(avg("total_ms"), 4)},
        }
        if two_stage:
            out["two_stage"] = {"bstar_suffixes": int(mstar), "induced_suffixes": int(n - mstar), "induction_ms": round(avg("other_ms"), 3),
                                "level_launches": int(phases[-1].reserved[6])}
        if len(ops) > 1:
            out["ops_ms"] = {k: round(v / K, 3) for k, v in op_ms.items()}
            if "ibwt" in ops:
                out["ibwt"] = {"walk_ms": round(ibwt_us[0] / K / 1e3, 3), "device_total_ms": round(ibwt_us[1] / K / 1e3, 3),
                               "walk_sector_GBps": round(64 * n / (ibwt_us[0] / K / 1e6) / 1e9, 1) if ibwt_us[0] else None}
        if latency:
            out.update(latency)           # one build incl. its exchange, nothing overlapped (ms_per_step above is the pipelined rate)
        if world > 1 and shard_state.stats:
            out["doubling"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in shard_state.stats.items()}
        if not args.no_cpu and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n), args.seed, args.workload)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
