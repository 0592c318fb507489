#!/bin/bash
# round 6: when do the transform's bytes leave?  host trace with the copier's marks + kernel trace of the device-resident call
ulimit -c 0
O=gpurun_out/r6o; mkdir -p $O
MSUFSORT_HIP_HOST_TRACE=1 timeout 600 python tools/gpu_host_text.py text 1073741823 1 fbwt > $O/host_text.txt 2>&1
grep -B2 -A80 "fbwt rep 0" $O/host_text.txt | grep -A80 "H2D done" | head -120
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace -o fbwt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_fbwt_dev.py text 1073741823 2 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r6o/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last build: from the last k_types on
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_types"))
t0 = int(rows[idx]["Start_Timestamp"])
out = open("gpurun_out/r6o/timeline.txt", "w")
for r in rows[idx:]:
    nm = r["Kernel_Name"][:40]
    if nm.startswith("k_bwt_region") or nm.startswith("k_find_row0") or nm.startswith("k_place") or nm.startswith("k_ind_check"):
        out.write(f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} {(int(r['End_Timestamp']) - t0) / 1e6:9.3f} ms  {nm} grid {r.get('Grid_Size_X', r.get('Grid_Size'))}\n")
out.close()
PY
head -100 gpurun_out/r6o/timeline.txt
