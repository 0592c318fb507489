#!/bin/bash
# round 6, final build: size sweep of random bytes (is the time per GiB flat?) and the crossovers of the two-stage build's default policy
ulimit -c 0
O=gpurun_out/r6sweeps; mkdir -p $O
python -c "from msufsort_amd import _lib; print('library build', _lib.lib().msufsort_hip_build_id().decode())" > $O/sizes.txt 2>/dev/null
timeout 900 python tools/gpu_r4_sizes.py 2>&1 | grep -v amdgpu >> $O/sizes.txt; tail -17 $O/sizes.txt | cut -c1-200
python -c "from msufsort_amd import _lib; print('library build', _lib.lib().msufsort_hip_build_id().decode())" > $O/two_stage_sweep.txt 2>/dev/null
timeout 900 python tools/gpu_two_stage_sweep.py text 32 48 64 80 96 128 256 2>&1 | grep -v amdgpu >> $O/two_stage_sweep.txt
timeout 900 python tools/gpu_two_stage_sweep.py dna 128 192 256 320 384 512 2>&1 | grep -v amdgpu >> $O/two_stage_sweep.txt
cat $O/two_stage_sweep.txt
