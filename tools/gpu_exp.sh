#!/bin/bash
ulimit -c 0
# timing experiment with an alternative build of the library (results invalid by construction)
LIBV=$1; shift
cp msufsort_amd/lib/libmsufsort_hip.so /tmp/lib_backup.so
cp msufsort_amd/lib/$LIBV msufsort_amd/lib/libmsufsort_hip.so
rm -rf /tmp/expraw; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/expraw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu "$@" > /tmp/exp.log 2>&1
cd $GRAFT_REPO_ROOT
cp /tmp/lib_backup.so msufsort_amd/lib/libmsufsort_hip.so
f=$(find /tmp/expraw -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    if 'validate' in r['Name'] or 'rocclr' in r['Name'] or 'at::native' in r['Name']: continue
    print(f"{r['Name'][:50]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f}")
PY
tail -2 /tmp/exp.log | cut -c1-200
