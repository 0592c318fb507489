#!/bin/bash
ulimit -c 0
O=gpurun_out/r6p; mkdir -p $O
MSUFSORT_HIP_HOST_TRACE=1 timeout 600 python tools/gpu_host_text.py text 1073741823 2 sa,fbwt > $O/host_text.txt 2>&1
grep -v "host trace" $O/host_text.txt; grep -E "built|copied out|induction: done|H2D" $O/host_text.txt | tail -12
timeout 600 python tools/gpu_host_text.py dna 1073741823 2 sa,fbwt > $O/host_dna.txt 2>&1; grep -v "host trace" $O/host_dna.txt
timeout 600 python tools/gpu_host_text.py random 1073741823 2 sa,fbwt > $O/host_random.txt 2>&1; grep -v "host trace" $O/host_random.txt
