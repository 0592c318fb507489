#!/usr/bin/env python3
"""Copies what tools/gpu_final_r3.sh left in gpurun_out/final/ into profiles/r03_* and rebuilds profiles/pmc_traffic.json (the
per-launch HBM traffic bench.py quotes) from the PMC pass: FETCH_SIZE x 2 (gfx950 counts 128-byte requests at 64 bytes,
MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes."""
import json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
NAMES = {
    "bench.json": "r03_bench.json", "bench_256MiB.json": "r03_bench_256MiB.json", "bench_2ranks_one_gpu_256MiB.json": "r03_bench_2ranks_one_gpu_256MiB.json",
    "bench_dna.json": "r03_bench_dna_1GiB.json", "bench_dna_tandem_256MiB.json": "r03_bench_dna_tandem_256MiB.json", "bench_fast2.json": "r03_bench_k_sort_fast2.json",
    "bench_text_sort_all.json": "r03_bench_text_sort_all.json", "big_inputs.log": "r03_big_inputs.log", "bits_phase_clocks.txt": "r03_bits_phase_clocks.txt",
    "kernel_stats_random.csv": "r03_kernel_stats_random_1GiB.csv", "kernel_stats_text.csv": "r03_kernel_stats_text_1GiB_sa_fbwt_ibwt_lcp.csv",
    "kernel_stats_dna.csv": "r03_kernel_stats_dna_1GiB.csv", "kernel_stats_tandem.csv": "r03_kernel_stats_dna_tandem_256MiB.csv",
    "microbench_exp_lds_rates.txt": "r03_microbench_exp_lds_rates.txt", "microbench_exp_lds_valu_overlap.txt": "r03_microbench_exp_lds_valu_overlap.txt",
    "microbench_exp_bits_phases.txt": "r03_microbench_exp_bits_phases.txt", "microbench_exp_bits_phases2.txt": "r03_microbench_exp_bits_phases2.txt",
    "pmc_sq_random.txt": "r03_pmc_sq_random_1GiB.txt", "pmc_sq_text.txt": "r03_pmc_sq_text_1GiB.txt", "pmc_traffic_random.txt": "r03_pmc_traffic_random_1GiB.txt",
    "pmc_traffic_text.txt": "r03_pmc_traffic_text_1GiB.txt", "pytest_gpu.log": "r03_pytest_gpu.log", "tandem_rounds.txt": "r03_tandem_rounds_256MiB.txt",
    "trace_text_two_stage.txt": "r03_trace_text_1GiB_two_stage.txt", "bits_sizes.txt": "r03_bucket_sort_size_sweep.txt", "bench_time.txt": "r03_bench_default_run_time.txt",
}
for a, b in NAMES.items():
    src = os.path.join(F, a)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, b))
    else:
        print("missing or empty:", a)
# the random-line micro-benchmark: independent reads (K = 8) + dependent chains (K = 1 binary prints both tables too)
parts = [open(os.path.join(F, f)).read() for f in ("microbench_exp_random_lines.txt", "microbench_exp_random_lines_k1.txt") if os.path.exists(os.path.join(F, f))]
if parts:
    open(os.path.join(P, "r03_microbench_exp_random_lines.txt"), "w").write("\n--- built with -DK=1 ---\n".join(parts))

LABEL = {"k_hist16<0>": "k_hist16", "k_scatter0<false>": "k_scatter0", "k_partition": "k_partition(level 1)", "k_sort_bits<1024": "bucket sort (LDS sorts of the two-byte buckets)"}
acc = {}
for line in open(os.path.join(F, "pmc_traffic_random.txt")):
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE) (?:void )?(.+?) calls (\d+) sum_KiB (\S+) per_call_KiB (\S+)", line.strip())
    if not m:
        continue
    for k, lab in LABEL.items():
        if m.group(2).startswith(k):
            acc[lab] = acc.get(lab, 0.0) + float(m.group(5)) * 1024 * (2 if m.group(1) == "FETCH_SIZE" else 1)
old = json.load(open(os.path.join(P, "pmc_traffic.json")))
old["kernels"] = {k: int(v) for k, v in acc.items()}
old["source"] = "profiles/r03_pmc_traffic_random_1GiB.txt"
json.dump(old, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(old["kernels"], indent=1))
