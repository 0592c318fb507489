timeout 1200 python tools/gpu_stress_big.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do python -m pytest tests -x -q -m gpu 2>&1 | tail -2; done
