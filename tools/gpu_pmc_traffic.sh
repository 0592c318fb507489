#!/bin/bash
# HBM traffic per launch (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel
# trace only), KiB per launch; gfx950 correction: FETCH_SIZE reads half of wide streaming reads.
# usage: tools/gpu_pmc_traffic.sh <out-file> [bench args...]
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcraw
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmcraw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu "$@" > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmcraw -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:44]
    agg[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(agg, key=lambda x: -agg[x])[:40]:
    if not k.startswith("void at::") and "rocclr" not in k:
        print(sys.argv[2], k, "calls", cnt[k], "sum_KiB", f"{agg[k]:.6g}", "per_call_KiB", f"{agg[k]/cnt[k]:.6g}")
PY
done | tee $OUT
