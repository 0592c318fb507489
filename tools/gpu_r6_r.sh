#!/bin/bash
# round 6: k_sort_tiny ranks by two keys per gather (GS_TINY2): parity subset + A/B
ulimit -c 0
O=gpurun_out/r6r; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep FAILED $O/pytest.log | head
for e in MSUFSORT_X=1 MSUFSORT_HIP_NO_TINY2=1; do
  for w in "text 1073741823 0" "text 1073741823 -1" "dna 1073741823 0" "text 268435456 -1"; do set -- $w
    echo "== $e $1 $2 two_stage=$3" >> $O/timings.txt
    env $e timeout 300 python tools/gpu_one.py $1 $2 $3 4 2>&1 | grep -E "build [123]|errors" >> $O/timings.txt
  done
done
paste - - - - - < $O/timings.txt | cut -c1-160
MSUFSORT_X=1 timeout 300 python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -E "round|two-stage" | cut -c1-200 > $O/rounds_tiny2.txt
MSUFSORT_HIP_NO_TINY2=1 timeout 300 python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -E "round|two-stage" | cut -c1-200 > $O/rounds_tiny1.txt
cat $O/rounds_tiny2.txt; cat $O/rounds_tiny1.txt
