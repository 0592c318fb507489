#!/bin/bash
# rocprofv3 kernel trace of the bench workload; summaries copied to gpurun_out/prof
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
SIZE=${1:-1073741823}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/raw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --size $SIZE --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof/kernel_stats.csv
rm -rf gpurun_out/prof/raw
cat gpurun_out/prof/kernel_stats.csv | head -40
tail -1 gpurun_out/prof/bench_under_prof.log | cut -c1-400
