#!/bin/bash
# round 6: the induction's rows carry dense numbers (pc_bits 2 .. 5) instead of plain bytes: parity + A/B
ulimit -c 0
O=gpurun_out/r6u; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "two_stage or bwt or host or fuzz or golden or edge" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep FAILED $O/pytest.log | head
for e in MSUFSORT_X=1 MSUFSORT_HIP_IND_PC_RAW=1; do
  for w in "dna 1073741823 0" "text 1073741823 0" "dna 402653184 0"; do set -- $w
    echo "== $e $1 $2" >> $O/timings.txt
    env $e timeout 300 python tools/gpu_one.py $1 $2 $3 4 2>&1 | grep -E "build [123]|errors" >> $O/timings.txt
  done
done
paste - - - - - < $O/timings.txt | cut -c1-160
