#!/bin/bash
# round 6: host-pointer text build - rows leave from the right-to-left pass already (B regions), ring for pieces from 1 MiB
ulimit -c 0
O=gpurun_out/r6m; mkdir -p $O
MSUFSORT_HIP_HOST_TRACE=1 timeout 600 python tools/gpu_host_text.py text 1073741823 2 sa > $O/trace_early_b.txt 2>&1
MSUFSORT_HIP_HOST_TRACE=1 MSUFSORT_HIP_NO_EARLY_B=1 timeout 600 python tools/gpu_host_text.py text 1073741823 2 sa > $O/trace_last_pass_only.txt 2>&1
timeout 600 python tools/gpu_host_text.py dna 1073741823 2 sa > $O/dna.txt 2>&1
tail -30 $O/trace_early_b.txt; tail -14 $O/trace_last_pass_only.txt; tail -5 $O/dna.txt
