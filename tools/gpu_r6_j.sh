#!/bin/bash
# round 6: k_ind_fused at 4 (default) / 5 / 6 waves per SIMD (register caps 128 / 96 / 80)
ulimit -c 0
O=gpurun_out/r6j; mkdir -p $O
for tag in default _i5 _i6; do
  lib=$PWD/msufsort_amd/lib/libmsufsort_hip_var$tag.so; [ $tag = default ] && lib=$PWD/msufsort_amd/lib/libmsufsort_hip.so
  for w in "text 1073741823" "dna 1073741823"; do set -- $w
    echo "== $tag $1" >> $O/ind_waves.txt
    MSUFSORT_HIP_LIB=$lib timeout 300 python tools/gpu_one.py $1 $2 0 3 2>&1 | grep -E "build [12]|errors" >> $O/ind_waves.txt
  done
done
cat $O/ind_waves.txt | paste - - - - | cut -c1-120
