#!/bin/bash
# induction tiles of 6144 / 8192 rows (variant builds -DIND_ITEMS=24 / 32) against 4096
ulimit -c 0
O=gpurun_out/r6items; mkdir -p $O; rm -f $O/t.txt
for v in default items8 items12; do
  lib=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip_var_$v.so; [ $v = default ] && lib=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip.so
  [ $v != default ] && ( MSUFSORT_HIP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage" 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/t.txt )
  for w in text dna; do
    MSUFSORT_HIP_LIB=$lib timeout 300 python tools/gpu_two_stage_only.py $w 1073741823 3 2>&1 | grep "induction ms" | tail -2 | sed "s/^/$v $w: /" >> $O/t.txt
  done
done
cat $O/t.txt
