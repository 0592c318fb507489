#!/bin/bash
# phase clocks of k_sort_bits on the headline input (prof build: -DBITS_PROF)
ulimit -c 0
O=gpurun_out/r6y; mkdir -p $O
MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip_prof_mid.so timeout 300 python tools/gpu_one.py random 1073741823 0 2 2>&1 | grep -E "bits prof|build" | cut -c1-400 > $O/bits_prof.txt
cat $O/bits_prof.txt
