#!/bin/bash
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --output-format csv -d /tmp/praw -- python3 $GRAFT_REPO_ROOT/tools/gpu_two_stage_only.py text 1073741823 2 > /tmp/p.log 2>&1
f=$(find /tmp/praw -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_ind_fused")]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]          # second build
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
g=[int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size",0)) for r in rows]
print("launches", len(d), "total ms", sum(d)/1e3)
for lo,hi in ((0,10),(10,30),(30,100),(100,300),(300,1000),(1000,1e9)):
    sel=[x for x in d if lo<=x<hi]
    print("  %5d..%-7s us: %4d launches, %8.2f ms" % (lo, hi if hi<1e9 else "", len(sel), sum(sel)/1e3))
top=sorted(zip(d,g), reverse=True)[:12]
print("longest:", ["%.0f us (grid %d)" % (a, b) for a,b in top])
t0=int(rows[0]["Start_Timestamp"]); t1=int(rows[-1]["End_Timestamp"])
print("span of the fused launches ms", (t1-t0)/1e6, "busy", sum(d)/1e3)
PY
