#!/bin/bash
# HBM traffic per kernel of ANY command (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel trace
# only), summed per kernel name; gfx950 correction (FETCH_SIZE counts 128-byte requests at 64 bytes) applied by the collector, not here.
# With PHASE_SPLIT=<kernel name prefix> the dispatches are also summed in two phases: before / from the first dispatch of that kernel
# (BASELINE config 5: the shard builds, then the distributed doubling, which starts with k_isa_from_slice).
# usage: tools/gpu_pmc_cmd.sh <out-file> <program> [args...]      (the program itself after --: never env / bash -c under rocprofv3 --pmc)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcraw
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmcraw -- "$@" > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmcraw -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr "${PHASE_SPLIT:-}" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
split = sys.argv[3]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
phase = [0.0, 0.0]; seen = False
for r in rows:
    name = r["Kernel_Name"]
    k = name.split("(")[0][:60]
    v = float(r["Counter_Value"])
    agg[k] += v; cnt[k] += 1
    if split:
        short = k.replace("void ", "")
        if short.startswith(split): seen = True
        if not (short.startswith("at::") or "rocclr" in short or "elementwise" in short): phase[1 if seen else 0] += v
for k in sorted(agg, key=lambda x: -agg[x])[:48]:
    if not k.startswith("void at::") and "rocclr" not in k:
        print(sys.argv[2], k, "calls", cnt[k], "sum_KiB", f"{agg[k]:.6g}", "per_call_KiB", f"{agg[k]/cnt[k]:.6g}")
if split:
    print(sys.argv[2], "PHASE before_" + split, "calls 1 sum_KiB", f"{phase[0]:.6g}", "per_call_KiB", f"{phase[0]:.6g}")
    print(sys.argv[2], "PHASE from_" + split, "calls 1 sum_KiB", f"{phase[1]:.6g}", "per_call_KiB", f"{phase[1]:.6g}")
PY
done | tee $OUT
