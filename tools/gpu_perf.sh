#!/bin/bash
ulimit -c 0
# perf iteration: 1 GiB bench (validated on device) + kernel-trace stats
mkdir -p gpurun_out/prof
python bench.py --steps 5 --warmup 2 --no-cpu ${BENCH_ARGS:-} 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_last.json
python3 -c "
import json; d=json.load(open('gpurun_out/bench_last.json')); print(d['config']['n'], d['value'],'MB/s', d['ms_per_step'],'ms', d['phases_ms'], 'valid', d.get('valid'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/raw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof/kernel_stats.csv
rm -rf gpurun_out/prof/raw
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/prof/kernel_stats.csv')))[:12]:
    if 'validate' in r['Name'] or 'rocclr' in r['Name'] or 'at::native' in r['Name']: continue
    print(f"{r['Name'][:50]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f}")
PY
