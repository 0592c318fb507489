#!/bin/bash
# round 4, first loop: new parity tests, size sweep, host-path legs of bench.py
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py -x -q -k "radix17 or host_entry or literal or class_api or generated" 2>&1 | tail -5 > gpurun_out/r4/pytest_a.txt
cat gpurun_out/r4/pytest_a.txt
timeout 900 python tools/gpu_r4_sizes.py > gpurun_out/r4/sizes.txt 2>&1; cat gpurun_out/r4/sizes.txt | grep -v amdgpu.ids
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu --no-configs > gpurun_out/r4/bench_host.json 2> gpurun_out/r4/bench_host.err; tail -c 3000 gpurun_out/r4/bench_host.json; tail -5 gpurun_out/r4/bench_host.err
