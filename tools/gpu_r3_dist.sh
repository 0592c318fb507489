#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --size 268435456 --no-cpu > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
tail -30 gpurun_out/bench_2rank.err | grep -v amdgpu.ids
python -m pytest tests -q -m gpu --deselect tests/test_gpu_big.py -k "not full_size" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
