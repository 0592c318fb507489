#!/usr/bin/env python3
"""Copies what tools/gpu_final_r6.sh left in gpurun_out/final6/ into profiles/r06_* and rebuilds profiles/pmc_traffic.json (the per-launch HBM
traffic bench.py quotes) from the PMC passes: FETCH_SIZE x 2 (gfx950 counts 128-byte requests at 64 bytes, MI355X_MICROARCH.md) + WRITE_SIZE,
KiB -> bytes - keyed by the build id of the library the passes ran with, so that bench.py drops the figures as soon as the library is rebuilt
from different sources."""
import json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, P = os.path.join(ROOT, "gpurun_out", "final6"), os.path.join(ROOT, "profiles")
for f in sorted(os.listdir(F)):
    src = os.path.join(F, f)
    if f == "build_id.txt" or not os.path.isfile(src):
        continue
    if os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, "r06_" + f))
    else:
        print("empty:", f)
build = open(os.path.join(F, "build_id.txt")).read().strip()
DOUBLING = "distributed prefix doubling (k_import_groups, k_chain_resolve, k_refill_rows, LDS sorts, k_emit_updates, k_apply_updates)"


def passes(name):
    """{kernel prefix: {"calls": c, "sum": FETCH x 2 + WRITE bytes}} of one PMC file"""
    acc = {}
    path = os.path.join(F, name)
    if not os.path.exists(path):
        return acc
    for line in open(path):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE) (?:void )?(.+?) calls (\d+) sum_KiB (\S+) per_call_KiB (\S+)", line.strip())
        if m:
            e = acc.setdefault(m.group(2), {"calls": int(m.group(3)), "sum": 0.0, "fetch_raw": 0.0})
            e["sum"] += float(m.group(4)) * 1024 * (2 if m.group(1) == "FETCH_SIZE" else 1)
            if m.group(1) == "FETCH_SIZE":
                e["fetch_raw"] += float(m.group(4)) * 1024
    return acc


def total(acc, *prefixes, field="sum"):
    return sum(e[field] for k, e in acc.items() if any(k.startswith(p) for p in prefixes))


rnd, rnd28 = passes("pmc_traffic_random.txt"), passes("pmc_traffic_random_256MiB.txt")
txt, wlk = passes("pmc_traffic_text_sa.txt"), passes("pmc_traffic_text_ibwt_lcp.txt")
c5, td = passes("pmc_traffic_cfg5.txt"), passes("pmc_traffic_dna_tandem_256MiB.txt")
n = (1 << 30) - 1
out = {"build_id": build,
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_pmc_cmd.sh, ONE build per pass); FETCH_SIZE x2 (gfx950 counts "
               "128-B requests at 64 B, MI355X_MICROARCH.md); KiB -> bytes; per launch, or per build for the phases that are many launches.  The x2 is calibrated for wide "
               "streaming reads only: for the gather-bound phases (k_ibwt_walk, k_lcp, induction, key rounds, the doubling of config 5) FETCH_SIZE raw is one 64-byte request "
               "per random access and the figure here - that x2 as prescribed - an upper bound (128 bytes per touch); `raw_fetch_plus_write` = FETCH x 1 + WRITE",
       "entries": {}}


def kernels_random(acc):
    return {"k_hist16": int(total(acc, "k_hist16<0>")), "k_scatter0": int(total(acc, "k_scatter0<false>")),
            "k_partition(level 1)": int(total(acc, "k_partition")),
            "bucket sort (LDS sorts of the two-byte buckets)": int(total(acc, "k_sort_bits"))}


if rnd:
    out["entries"]["random"] = {"n": n, "source": "profiles/r06_pmc_traffic_random.txt", "kernels": kernels_random(rnd)}
if rnd28:
    out["entries"]["random@268435456"] = {"n": 1 << 28, "source": "profiles/r06_pmc_traffic_random_256MiB.txt", "kernels": kernels_random(rnd28)}
if txt:
    key_p = ("k_sort_mid", "k_sort_tiny", "k_count", "k_refill", "k_carry_copy")
    ent = {"k_hist16": int(total(txt, "k_hist16<0>")), "k_scatter0": int(total(txt, "k_scatter0<false>")),
           "induction (k_ind_fused + k_ind_small)": int(total(txt, "k_ind_fused", "k_ind_small")),
           # everything the rounds behind round 0 run: LDS sorts with their gathers, partition levels, refills (the round-0 share of the
           # sorts cannot be told apart in a per-kernel sum: an upper bound for the key rounds)
           "key rounds (k_refill + k_partition levels + LDS sorts)": int(total(txt, *key_p))}
    out["entries"]["text"] = {"n": n, "source": "profiles/r06_pmc_traffic_text_sa.txt", "kernels": ent,
                              "raw_fetch_plus_write": {"induction (k_ind_fused + k_ind_small)": int(total(txt, "k_ind_fused", "k_ind_small") - total(txt, "k_ind_fused", "k_ind_small", field="fetch_raw")),
                                                       "key rounds (k_refill + k_partition levels + LDS sorts)": int(total(txt, *key_p) - total(txt, *key_p, field="fetch_raw"))}}
dna = passes("pmc_traffic_dna.txt")
if dna:
    key_p = ("k_sort_mid", "k_sort_tiny", "k_count", "k_refill", "k_carry_copy", "k_sort_fast2")
    out["entries"]["dna"] = {"n": n, "source": "profiles/r06_pmc_traffic_dna.txt",
                             "kernels": {"k_hist16": int(total(dna, "k_hist16<0>")), "k_scatter0": int(total(dna, "k_scatter0<false>")),
                                         "induction (k_ind_fused + k_ind_small)": int(total(dna, "k_ind_fused", "k_ind_small")),
                                         "key rounds (k_refill + k_partition levels + LDS sorts)": int(total(dna, *key_p))},
                             "raw_fetch_plus_write": {"induction (k_ind_fused + k_ind_small)": int(total(dna, "k_ind_fused", "k_ind_small") - total(dna, "k_ind_fused", "k_ind_small", field="fetch_raw"))}}
if wlk and "text" in out["entries"]:
    out["entries"]["text"]["kernels"]["k_ibwt_walk"] = int(total(wlk, "k_ibwt_walk"))
    out["entries"]["text"]["kernels"]["k_lcp"] = int(total(wlk, "k_lcp"))
if c5:
    ph = {k: e for k, e in c5.items() if k.startswith("PHASE ")}
    dbl = sum(e["sum"] for k, e in ph.items() if k.startswith("PHASE from_"))
    dbl_raw = sum(e["sum"] - e["fetch_raw"] for k, e in ph.items() if k.startswith("PHASE from_"))
    shards = sum(e["sum"] for k, e in ph.items() if k.startswith("PHASE before_"))
    out["entries"]["dna_tandem"] = {"n": 1 << 33, "source": "profiles/r06_pmc_traffic_cfg5.txt (tools/gpu_cfg5.py 33 1 32: dispatches from the first k_isa_from_slice<true> on = the doubling phase)",
                                    "kernels": {DOUBLING: int(dbl), "shard builds (round 0 + three key rounds of the 32 shards)": int(shards)},
                                    "raw_fetch_plus_write": {DOUBLING: int(dbl_raw)}}
if td:
    out["entries"]["dna_tandem@268435456"] = {"n": 1 << 28, "source": "profiles/r06_pmc_traffic_dna_tandem_256MiB.txt",
                                              "kernels": {k: int(e["sum"]) for k, e in sorted(td.items(), key=lambda kv: -kv[1]["sum"])[:12] if not k.startswith("PHASE")}}
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
