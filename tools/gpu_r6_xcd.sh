#!/bin/bash
# experiment: the induction's tiles mapped to XCDs by eighths of a round (MSUFSORT_HIP_IND_XCD=1, variant library)
ulimit -c 0
O=gpurun_out/r6xcd; mkdir -p $O; rm -f $O/t.txt
lib=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip_var_xcd.so
MSUFSORT_HIP_IND_XCD=1 MSUFSORT_HIP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage" 2>&1 | tail -1 >> $O/t.txt
for x in 0 1; do for w in text dna; do
  MSUFSORT_HIP_IND_XCD=$x MSUFSORT_HIP_LIB=$lib timeout 300 python tools/gpu_two_stage_only.py $w 1073741823 3 2>&1 | grep -E "induction ms|two-stage" | tail -3 | cut -c1-230 | sed "s/^/xcd=$x $w: /" >> $O/t.txt
done; done
cat $O/t.txt
