#!/bin/bash
ulimit -c 0
O=gpurun_out/r6t; mkdir -p $O
for lb in 3 100; do
  for w in "dna 1073741823 0" "text 1073741823 0"; do set -- $w
    echo "== IND_LB=$lb $1" >> $O/timings.txt
    MSUFSORT_HIP_IND_LB=$lb timeout 300 python tools/gpu_two_stage_only.py $1 $2 2>&1 | tail -2 >> $O/timings.txt
  done
done
cat $O/timings.txt
