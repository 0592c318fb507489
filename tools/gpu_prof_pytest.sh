#!/bin/bash
# kernel-trace stats of one pytest selection: tools/gpu_prof_pytest.sh <out-name> <pytest args...>
O=gpurun_out/prof; mkdir -p $O
NAME=$1; shift
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- python3 -m pytest --rootdir=$GRAFT_REPO_ROOT $GRAFT_REPO_ROOT/tests "$@" > /tmp/p.log 2>&1; grep -E "passed|failed|n=" /tmp/p.log | tail -5; find /tmp/praw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/$NAME.csv )
python3 - <<PY
import csv
rows=list(csv.reader(open("$O/$NAME.csv")))
for r in rows[1:24]:
    print(r[0][:60].ljust(60), r[1].rjust(6), "%10.3f ms total" % (float(r[2])/1e6), "%8.3f ms avg" % (float(r[3])/1e6), r[4])
PY
