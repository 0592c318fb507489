#!/bin/bash
mkdir -p gpurun_out/r4
python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -v amdgpu.ids | grep -i "two-stage\|induction\|total" | tail -5
MSUFSORT_HIP_IND_GRID=100000 python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -i "two-stage:" | tail -2
MSUFSORT_HIP_IND_GRID=1024 python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -i "two-stage:" | tail -2
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from msufsort_amd import gen
gen.random_bytes((1 << 30) - 1, 12345).tofile("/dev/shm/r.bin")
PY
MSUFSORT_HIP_HOST_TRACE=1 build/host_bench /dev/shm/r.bin 2 2>&1 | grep -v "shard built\|slice copy" | tail -40
rm -f /dev/shm/r.bin
