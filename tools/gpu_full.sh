#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_perf.sh 2>&1 | tail -9
echo "=== cfg3/4 text 2^30-1"; timeout 1200 python tools/gpu_configs.py text 1073741823 ref 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cfg3_text.log | tail -12
