#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --size 268435456 --no-cpu 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 2>&1 | tail -2 | tee gpurun_out/bench_r01_first.json
