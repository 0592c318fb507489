#!/bin/bash
# inverse-BWT walk time of library variants (msufsort_amd/lib/libmsufsort_hip_var<tag>.so; "base" = product): tools/gpu_variants_ibwt2.sh base _x ...
for tag in "$@"; do
  if [ "$tag" = base ]; then unset MSUFSORT_HIP_LIB; else export MSUFSORT_HIP_LIB=$PWD/msufsort_amd/lib/libmsufsort_hip_var$tag.so; fi
  echo "== $tag"; python tools/gpu_ibwt_time.py 2>&1 | tail -2
done
