#!/bin/bash
ulimit -c 0
O=gpurun_out/r6x; mkdir -p $O
R=$GRAFT_REPO_ROOT
for w in dna text; do
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- python3 $R/tools/gpu_two_stage_only.py $w 1073741823 3 > /tmp/p.log 2>&1; find /tmp/praw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/$O/$w.csv )
python3 - $O/$w.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:]:
    if any(k in r[0] for k in ("k_types", "k_hist16", "k_maxrun", "k_reduce16", "k_place", "k_ind_fused")):
        print("  %10.3f ms/build %7.1f launches  avg %9.1f us  %s" % (float(r[2]) / 1e6 / 3, float(r[1]) / 3, float(r[3]) / 1e3, r[0][:60]))
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage" 2>&1 | tail -2
