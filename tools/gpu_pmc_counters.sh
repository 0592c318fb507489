#!/bin/bash
# any PMC counters (one pass) over one bench.py line, summed per kernel: tools/gpu_pmc_counters.sh "<counters>" <out-file> [bench args...]
ulimit -c 0
CTRS="$1"; OUT=$GRAFT_REPO_ROOT/$2; shift 2
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcraw
timeout 400 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmcraw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-configs "$@" > /tmp/pmc.log 2>&1
f=$(find /tmp/pmcraw -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then tail -5 /tmp/pmc.log; exit 1; fi
python3 - "$f" <<'PY' | tee $OUT
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:34]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
print("kernel".ljust(36), " ".join(n[-22:].rjust(22) for n in names))
for k in sorted(agg, key=lambda x: -sum(agg[x].values()))[:14]:
    if k.startswith("void at::") or "rocclr" in k: continue
    print(k.ljust(36), " ".join(("%.4g" % agg[k][n]).rjust(22) for n in names))
PY
