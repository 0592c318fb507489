#!/bin/bash
mkdir -p gpurun_out/r4
MSUFSORT_HIP_HOST_TRACE=1 timeout 300 python tools/gpu_host_fresh.py 1073741823 random 3 > gpurun_out/r4/host_trace_random.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4/host_trace_random.txt | tail -75
timeout 600 python tools/gpu_r4_sizes.py 270 285 296 320 400 512 1120 1180 > gpurun_out/r4/sizes_b.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4/sizes_b.txt
