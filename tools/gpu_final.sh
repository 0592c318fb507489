#!/bin/bash
# End-of-round snapshot: parity tests, bench (with CPU baseline), kernel-trace stats, PMC traffic, configs 2-4,
# micro-benchmarks.  Everything lands in gpurun_out/final/.
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
python bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench.json; cat $O/bench.json | cut -c1-400
python bench.py --steps 5 --warmup 2 --size 268435456 --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_256MiB.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fraw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fraw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /tmp/f.log 2>&1; find /tmp/fraw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/kernel_stats.csv )
bash tools/gpu_pmc_traffic.sh > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt
timeout 600 python tools/gpu_configs.py text 1073741823 ref > $O/cfg3_cfg4_text.log 2>&1; grep -E "SA:|BWT|LCP|RESULT|reference" $O/cfg3_cfg4_text.log
timeout 300 python tools/gpu_configs.py dna 1073741823 > $O/dna_1GiB.log 2>&1; grep -E "SA:|RESULT" $O/dna_1GiB.log
( timeout 100 tools/microbench/bin/exp_hist 1073741823 0; timeout 100 tools/microbench/bin/exp_hist 1073741823 2; timeout 100 tools/microbench/bin/exp_write ) > $O/microbench.txt 2>&1
