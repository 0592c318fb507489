#!/bin/bash
# round 3: full GPU test suite, then the driver's bench line (headline + configs 3/4 + CPU legs), then --gpus 2 on one device
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
( time python bench.py --steps 10 --warmup 3 ) 2> gpurun_out/bench_time.log | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_last.json
tail -4 gpurun_out/bench_time.log
python3 -c "
import json; d=json.load(open('gpurun_out/bench_last.json')); print(d['value'],'MB/s', d['ms_per_step'],'ms', d['phases_ms'], 'valid', d.get('valid'), d.get('valid_against'))
print('cpu', d.get('cpu_baseline'))
print('cfg3', json.dumps(d.get('configs',{}).get('cfg3'))[:1500])
print('cfg4', json.dumps(d.get('configs',{}).get('cfg4'))[:800])"
MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --size 268435456 --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_2rank.json
python3 -c "
import json; d=json.load(open('gpurun_out/bench_2rank.json')); print('2 ranks:', d['n_gpus'], d['value'], d['ms_per_step'], d['valid'], d['valid_against'], d['latency_ms'], d['per_rank'])"
