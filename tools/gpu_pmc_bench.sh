#!/bin/bash
# PMC counters over bench.py (headline only); usage: gpu_pmc_bench.sh "<counters>" <kernel-name filter> [bench args...]
ulimit -c 0
CTRS="$1"; FILT="$2"; shift; shift
rm -rf /tmp/pmcraw; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmcraw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-configs "$@" > /tmp/pmc.log 2>&1
f=$(find /tmp/pmcraw -name "*counter_collection.csv" | head -1)
python3 - "$f" "$FILT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if sys.argv[2] in k: print(k, {a: f"{b:.4g}" for a, b in d.items()})
PY
