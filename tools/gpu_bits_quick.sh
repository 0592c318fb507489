#!/bin/bash
# quick check of the bucket sort: a parity subset, then headline + config 2 timings (+ optional SQ counters with PMC=1)
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "generated_golden or vs_oracle or edge or forced or bucket_sort or large_random or retry" > gpurun_out/pytest_bits.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bits.log
tail -3 gpurun_out/pytest_bits.log
for sz in 1073741823 268435456; do
  python bench.py --steps 10 --warmup 3 --size $sz --no-cpu --no-configs 2>gpurun_out/bench_bits.err | tail -1 > gpurun_out/bench_bits_$sz.json
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/bench_bits_$sz.json')); print($sz, d['value'],'MB/s', d['ms_per_step'],'ms', d['phases_ms'], 'valid', d.get('valid'), d.get('valid_against')[:30])" || tail -5 gpurun_out/bench_bits.err
done
if [ -n "$PMC" ]; then bash tools/gpu_pmc_sq.sh gpurun_out/pmc_sq_bits.txt --no-configs | grep -E "kernel|k_sort_bits"; fi
