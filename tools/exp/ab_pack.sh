#!/bin/bash
ulimit -c 0
for w in dna text; do for np in 0 1; do
  if [ $np = 1 ]; then export MSUFSORT_HIP_NO_PACK=1; else unset MSUFSORT_HIP_NO_PACK; fi
  echo "== $w NO_PACK=$np"; timeout 300 python tools/gpu_configs.py $w ${1:-1073741823} 2>&1 | grep -E "SA:|checker|RESULT"
done; done
