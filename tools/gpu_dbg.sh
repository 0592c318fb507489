#!/bin/bash
ulimit -c 0
for n in 67108864 268435456; do
  echo "=== text $n"; timeout 600 python tools/gpu_configs.py text $n ref 2>&1 | grep -v amdgpu.ids | tail -9
done
echo "=== dna"; timeout 600 python tools/gpu_configs.py dna 268435456 ref 2>&1 | grep -v amdgpu.ids | tail -9
