#!/bin/bash
# round 6: the new parity tests; phase clocks of the tile sort on the 1 GiB text
ulimit -c 0
O=gpurun_out/r6g; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -k "tiles_match or preceding or reuse_the_plan or sub_shards or key1 or histogram" ) > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log; grep FAILED $O/pytest_new.log | head
MSUFSORT_HIP_LIB=$PWD/msufsort_amd/lib/libmsufsort_hip_prof_mid.so timeout 300 python tools/gpu_one.py text 1073741823 0 1 > $O/mid_prof_text_tiles.txt 2>&1
grep "class A" $O/mid_prof_text_tiles.txt | head -6 | cut -c1-250
