#!/bin/bash
# kernel-trace stats of a python script: tools/gpu_prof_py.sh <out-name> <script> <args...>
O=gpurun_out/prof; mkdir -p $O
NAME=$1; shift
SCRIPT=$1; shift
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > /tmp/p.log 2>&1; grep -v "round " /tmp/p.log | tail -4 | cut -c1-300; find /tmp/praw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/$NAME.csv )
python3 - <<PY
import csv
rows=list(csv.reader(open("$O/$NAME.csv")))
for r in rows[1:24]:
    print(r[0][:60].ljust(60), r[1].rjust(6), "%10.3f ms total" % (float(r[2])/1e6), "%8.3f ms avg" % (float(r[3])/1e6), r[4])
PY
