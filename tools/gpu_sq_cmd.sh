#!/bin/bash
# What the kernels of ANY command wait for: one --pmc pass of SQ counters (kernel trace only).  usage: tools/gpu_sq_cmd.sh <out-file> <program> [args...]
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcraw
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcraw -- "$@" > /tmp/pmc.log 2>&1
f=$(find /tmp/pmcraw -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:52]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]
print("kernel".ljust(54), "launches", " ".join(n[3:].rjust(16) for n in names), "  wait_any/wave_cycles  active/wave_cycles")
for k in sorted(agg, key=lambda x: -agg[x]["SQ_WAVE_CYCLES"])[:14]:
    if k.startswith("void at::") or "rocclr" in k: continue
    a = agg[k]; wc = max(a["SQ_WAVE_CYCLES"], 1)
    print(k.ljust(54), str(calls[k]).rjust(8), " ".join(("%.4g" % a[n]).rjust(16) for n in names), "  %.3f  %.3f" % (a["SQ_WAIT_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc))
PY
