#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python tools/gpu_r4_sizes.py > gpurun_out/r4/sizes.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4/sizes.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_dist.py -x -q --durations=8 2>&1 | tail -14
