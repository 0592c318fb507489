#!/bin/bash
# round 6: look-back window of the induction: LB lanes per byte value (8 / 16 / 32 / 64)
ulimit -c 0
O=gpurun_out/r6s; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep FAILED $O/pytest.log | head
for lb in 3 4 5 6; do
  for w in "dna 1073741823 0" "text 1073741823 0"; do set -- $w
    echo "== LB=2^$lb $1" >> $O/timings.txt
    MSUFSORT_HIP_IND_LB=$lb timeout 300 python tools/gpu_two_stage_only.py $1 $2 2>&1 | tail -2 >> $O/timings.txt
  done
done
cat $O/timings.txt
