#!/usr/bin/env python3
"""Copies what tools/gpu_final_r5.sh left in gpurun_out/final/ into profiles/r05_* and rebuilds profiles/pmc_traffic.json (the
per-launch HBM traffic bench.py quotes) from the PMC passes: FETCH_SIZE x 2 (gfx950 counts 128-byte requests at 64 bytes,
MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes - keyed by the build id of the library the passes ran with, so that bench.py
drops the figures as soon as the library is rebuilt from different sources."""
import json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
# only what tools/gpu_final_r5.sh writes (gpurun_out/final/ also holds earlier rounds' files)
FILES = ["pytest_gpu.log", "bench.json", "bench_time.txt", "bench_256MiB.json", "bench_text_sort_all.json", "bench_dna.json", "bench_dna_tandem_256MiB.json",
         "bench_2ranks_one_gpu_256MiB.json", "bench_2ranks_one_gpu_64MiB.json", "bench_2ranks_one_gpu_text_256MiB_two_stage_sharded.json", "sizes.txt",
         "kernel_stats_random.txt", "kernel_stats_random.csv", "kernel_stats_text.txt", "kernel_stats_text.csv", "kernel_stats_2GiB.txt", "kernel_stats_2GiB.csv",
         "pmc_traffic_random.txt", "pmc_traffic_text_sa.txt", "pmc_traffic_text_ibwt_lcp.txt", "pmc_sq_text.txt", "pmc_sq_random.txt",
         "induction_level_durations.txt", "host_trace_random.txt", "host_trace_text.txt", "microbench_host_xfer.txt", "microbench_h2d_fresh.txt",
         "text_rounds.txt", "tandem_rounds.txt", "stress_350MB.txt", "pmc_traffic_random_2GiB.txt",
         "bench_4ranks_one_gpu_int64_dna_tandem_16MiB.json", "bench_2ranks_one_gpu_256MiB_sa_fbwt.json", "kernel_stats_dna.txt", "kernel_stats_dna.csv",
         "pmc_traffic_text_sa_key1_off.txt", "key1_ab.txt", "trace_text_two_stage.txt", "two_stage_sweep.txt"]
for f in FILES:
    src = os.path.join(F, f)
    if os.path.isfile(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, "r05_" + f))
    else:
        print("missing or empty:", f)
build = open(os.path.join(F, "build_id.txt")).read().strip()


def passes(name):
    """{kernel prefix: {"calls": c, "bytes_per_call": FETCH x 2 + WRITE}} of one PMC file"""
    acc = {}
    path = os.path.join(F, name)
    if not os.path.exists(path):
        return acc
    for line in open(path):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE) (?:void )?(.+?) calls (\d+) sum_KiB (\S+) per_call_KiB (\S+)", line.strip())
        if m:
            e = acc.setdefault(m.group(2), {"calls": int(m.group(3)), "sum": 0.0})
            e["sum"] += float(m.group(4)) * 1024 * (2 if m.group(1) == "FETCH_SIZE" else 1)
    return acc


def total(acc, *prefixes):
    return sum(e["sum"] for k, e in acc.items() if any(k.startswith(p) for p in prefixes))


rnd = passes("pmc_traffic_random.txt")
txt = passes("pmc_traffic_text_sa.txt")
wlk = passes("pmc_traffic_text_ibwt_lcp.txt")
n = (1 << 30) - 1
out = {"build_id": build,
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_pmc_traffic.sh: bench.py --steps 1 --warmup 0, ONE build); "
               "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md); units KiB -> bytes; per launch, or per build for the phases that are many launches.  "
               "The x2 is calibrated for wide streaming reads only (the guide says so): for the gather-bound kernels of the text entry (k_ibwt_walk, k_lcp, induction, key rounds) "
               "FETCH_SIZE raw is one 64-byte request per random access (k_ibwt_walk: 68.7 GB raw = 1.07 G requests = n hops), the figure here is that x2 as prescribed - "
               "read it as an upper bound (128 bytes per touch), raw = half of it",
       "entries": {}}
if rnd:
    out["entries"]["random"] = {"n": n, "source": "profiles/r05_pmc_traffic_random.txt", "kernels": {
        "k_hist16": int(total(rnd, "k_hist16<0>")), "k_scatter0": int(total(rnd, "k_scatter0<false>")),
        "k_partition(level 1)": int(total(rnd, "k_partition<256", "k_partition<512", "k_partition")),
        "bucket sort (LDS sorts of the two-byte buckets)": int(total(rnd, "k_sort_bits<1024"))}}
if txt:
    ent = {"k_hist16": int(total(txt, "k_hist16<0>")), "k_scatter0": int(total(txt, "k_scatter0<false>")),
           "induction (k_ind_fused + k_ind_small)": int(total(txt, "k_ind_fused", "k_ind_small")),
           # everything the rounds behind round 0 run: LDS sorts with their gathers, partition levels, refills (the round-0 share of the
           # sorts cannot be told apart in a per-kernel sum: this figure is an upper bound for the key rounds)
           "key rounds (k_refill + k_partition levels + LDS sorts)": int(total(txt, "k_sort_mid", "k_sort_tiny", "k_count", "k_refill", "k_carry_copy"))}
    out["entries"]["text"] = {"n": n, "source": "profiles/r05_pmc_traffic_text_sa.txt", "kernels": ent}
if wlk and "text" in out["entries"]:
    out["entries"]["text"]["kernels"]["k_ibwt_walk"] = int(total(wlk, "k_ibwt_walk"))
    out["entries"]["text"]["kernels"]["k_lcp"] = int(total(wlk, "k_lcp"))
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
