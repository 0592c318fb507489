#!/bin/bash
# round 6: forward transform riding on the two-stage build (bytes per final region, second stream) + host-pointer streaming of them
ulimit -c 0
O=gpurun_out/r6n; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage or host or bwt" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep FAILED $O/pytest.log | head
MSUFSORT_HIP_HOST_TRACE=1 timeout 600 python tools/gpu_host_text.py text 1073741823 2 sa,fbwt > $O/host_text.txt 2>&1
MSUFSORT_HIP_NO_BWT_RIDE=1 timeout 600 python tools/gpu_host_text.py text 1073741823 2 fbwt > $O/host_text_no_ride.txt 2>&1
timeout 600 python tools/gpu_host_text.py dna 1073741823 2 sa,fbwt > $O/host_dna.txt 2>&1
grep -v "host trace" $O/host_text.txt; grep -A12 "fbwt rep 1" $O/host_text.txt | grep "host trace"; grep -v "host trace" $O/host_text_no_ride.txt; cat $O/host_dna.txt
for e in MSUFSORT_X=1 MSUFSORT_HIP_NO_BWT_RIDE=1; do
  echo "== device fbwt $e" >> $O/dev.txt
  for w in text dna; do
  env $e timeout 300 python tools/gpu_fbwt_dev.py $w 1073741823 4 >> $O/dev.txt 2>&1
  done
done
cat $O/dev.txt
