#!/bin/bash
# per-dispatch kernel trace of a python script (last N dispatches): tools/gpu_trace_py.sh <out-name> <tail-count> <script> <args...>
O=gpurun_out/prof; mkdir -p $O
NAME=$1; shift
CNT=$1; shift
SCRIPT=$1; shift
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --output-format csv -d /tmp/praw -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > /tmp/p.log 2>&1; grep "round \|two-stage" /tmp/p.log | tail -16; find /tmp/praw -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} /tmp/trace.csv )
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-$CNT:]
t0=int(rows[0]["Start_Timestamp"])
out=open("$O/$NAME.txt","w")
for r in rows:
    name=r["Kernel_Name"][:70]
    if name.startswith("k_ind_") and not name.startswith("k_ind_small"): continue
    line="%9.3f ms  +%8.3f ms  %s" % ((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, name)
    out.write(line+"\n")
out.close()
PY
grep -v "k_zero_idx\|k_tiles\|k_segscan\|fillBuffer\|copyBuffer\|k_copy_idx" $O/$NAME.txt | tail -120
