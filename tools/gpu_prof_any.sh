#!/bin/bash
# kernel-trace stats for tools/gpu_configs.py <workload> <n>
ulimit -c 0
rm -rf /tmp/praw; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- python3 $GRAFT_REPO_ROOT/tools/gpu_configs.py "$@" > /tmp/p.log 2>&1
cd $GRAFT_REPO_ROOT
grep -E "SA:|RESULT|inverse|BWT from|LCP" /tmp/p.log
f=$(find /tmp/praw -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out/prof; cp $f gpurun_out/prof/kernel_stats_$1.csv
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f"{r['Name'][:56]:56s} calls={r['Calls']:>4s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:10.1f}")
PY
