#!/bin/bash
# A/B runs of alternative builds of the library (build/lib_NAME.so, selected with MSUFSORT_HIP_LIB): tools/gpu_variants.sh NAME...
for v in "$@"; do
  export MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$v.so
  echo "== $v: text 1GiB / tandem 256MiB / dna 1GiB"
  python tools/gpu_verbose.py text 1073741823 2>&1 | tail -1 | cut -c1-30
  python tools/gpu_verbose.py dna_tandem 268435456 2>&1 | tail -1 | cut -c1-30
  python tools/gpu_verbose.py dna 1073741823 2>&1 | tail -1 | cut -c1-30
done
