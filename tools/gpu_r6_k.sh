#!/bin/bash
# round 6: robustness of the new kernels (class-A tiles, characters picked up by the sorts, overflow flag of the tied-row lists): repeated builds
# compared bit for bit, adversarial and pathological inputs against the reference / the on-device checker, a corpus of real files
ulimit -c 0
O=gpurun_out/r6k; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
( time timeout 900 python tools/gpu_determinism.py ) > $O/determinism.txt 2>&1; tail -8 $O/determinism.txt
( time timeout 900 python tools/gpu_stress.py ) > $O/stress.txt 2>&1; tail -6 $O/stress.txt
( time timeout 900 python tools/gpu_patho.py ) > $O/patho.txt 2>&1; tail -6 $O/patho.txt
( time timeout 900 python tools/gpu_real_corpus.py ) > $O/real_corpus.txt 2>&1; tail -4 $O/real_corpus.txt
( time timeout 1200 python tools/gpu_stress_big.py ) > $O/stress_big.txt 2>&1; tail -5 $O/stress_big.txt
