#!/bin/bash
# kernel-trace stats of bench.py with the given arguments: tools/gpu_prof_bench.sh <out-name> <bench args...>
O=gpurun_out/prof; mkdir -p $O
NAME=$1; shift
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu "$@" > /tmp/p.log 2>&1; tail -1 /tmp/p.log | cut -c1-300; find /tmp/praw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/$NAME.csv )
python3 - <<PY
import csv
rows=list(csv.reader(open("$O/$NAME.csv")))
for r in rows[1:16]:
    print(r[0][:60].ljust(60), r[1].rjust(5), "%10.3f ms total" % (float(r[2])/1e6), "%8.3f ms avg" % (float(r[3])/1e6), r[4])
PY
