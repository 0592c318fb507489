#!/bin/bash
ulimit -c 0
# PMC counters for selected kernels (separate passes, no tracing domains other than kernel-trace)
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
SIZE=${SIZE:-268435456}
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/raw$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --size $SIZE --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/pmc/log$i.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc/raw$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if any(s in k for s in ("sort_fast", "partition", "scatter0", "hist16")):
        print(k, {a: f"{b:.4g}" for a, b in d.items()})
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc/raw$i
done
