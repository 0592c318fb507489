#!/bin/bash
# round 6: class-B tiles (k_sort_mid_btiles): parity subset, timings at 4 / 3 / 2 workgroups per CU against the one-segment instance
ulimit -c 0
O=gpurun_out/r6l; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep FAILED $O/pytest.log | head
for tag in default _b3 _b2; do
  lib=$PWD/msufsort_amd/lib/libmsufsort_hip_var$tag.so; [ $tag = default ] && lib=$PWD/msufsort_amd/lib/libmsufsort_hip.so
  for w in "text 1073741823" "dna 1073741823" "dna_tandem 268435456"; do set -- $w
    echo "== btiles$tag $1" >> $O/timings.txt
    MSUFSORT_HIP_LIB=$lib timeout 300 python tools/gpu_one.py $1 $2 0 3 2>&1 | grep -E "build [12]|errors" >> $O/timings.txt
  done
done
for w in "text 1073741823" "dna_tandem 268435456"; do set -- $w
  echo "== single $1" >> $O/timings.txt
  MSUFSORT_HIP_MID_SINGLE=1 timeout 300 python tools/gpu_one.py $1 $2 0 3 2>&1 | grep -E "build [12]|errors" >> $O/timings.txt
done
paste - - - - < $O/timings.txt | cut -c1-120
