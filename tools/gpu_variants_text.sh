#!/bin/bash
# library variants (make -C msufsort_amd/csrc variant VAR_FLAGS=... VAR_TAG=_x) on one workload, round by round:
#   tools/gpu_variants_text.sh <workload> <n> <two_stage> <tag> [<tag> ...]      ("base" = the product library)
W=$1; N=$2; TS=$3; shift 3
for tag in "$@"; do
  if [ "$tag" = base ]; then unset MSUFSORT_HIP_LIB; else export MSUFSORT_HIP_LIB=$PWD/msufsort_amd/lib/libmsufsort_hip_var$tag.so; fi
  echo "== $tag"
  python tools/gpu_verbose_any.py $W $N $TS 2>&1 | grep -E "round [0-4] mode|two-stage:|errors" | sed 's/.*levels + sorts/   levels + sorts/; s/\[msufsort_hip\] //'
done
