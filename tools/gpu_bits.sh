#!/bin/bash
# k_sort_bits vs k_sort_fast2: parity subset, then the headline and config 2 with either bucket sort
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/pytest_bits.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bits.log
tail -5 gpurun_out/pytest_bits.log
for v in bits fast2; do
  for sz in 1073741823 268435456; do
    MSUFSORT_HIP_BUCKET_SORT=$v python bench.py --steps 10 --warmup 3 --size $sz --no-cpu --no-configs 2>gpurun_out/bench_$v.err | tail -1 > gpurun_out/bench_${v}_$sz.json
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/bench_${v}_$sz.json')); print('$v', $sz, d['value'],'MB/s', d['ms_per_step'],'ms', d['phases_ms'], 'valid', d.get('valid'), d.get('valid_against')[:30])" || tail -5 gpurun_out/bench_$v.err
  done
done
