#!/bin/bash
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py -x -q -k "radix17 or two_stage or golden" 2>&1 | tail -3
timeout 900 python tools/gpu_r4_sizes.py > gpurun_out/r4/sizes.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4/sizes.txt
python -m pytest tests/test_gpu_big.py -x -q -k "class_limits or hist17" 2>&1 | tail -3
