#!/bin/bash
# kernel times of the inverse BWT front end for alternative builds of the library: tools/gpu_ibwt_variants.sh <lib> [<lib> ...] ("-" = the product library)
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  [ "$L" = "-" ] && LIB="" || LIB=$GRAFT_REPO_ROOT/$L
  D=/tmp/pr$RANDOM
  MSUFSORT_HIP_LIB=$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/gpu_ibwt_time.py > /tmp/iv.log 2>&1
  echo "== lib=$L: $(grep 'inverse device' /tmp/iv.log | tail -1)"
  python3 - <<PY
import csv, glob
f = glob.glob("$D/*/*kernel_stats.csv")[0]
for r in csv.reader(open(f)):
    if "k_ibwt" in r[0]:
        print("   %-28s calls %3s avg %8.3f ms" % (r[0][:28], r[1], float(r[3]) / 1e6))
PY
done
