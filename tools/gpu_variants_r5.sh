#!/bin/bash
# device time of the text / DNA suffix-array builds for alternative builds of the library: tools/gpu_variants_r5.sh <lib|default>...
cd "$(dirname "$0")/.."
for lib in "$@"; do
    if [ "$lib" = default ]; then unset MSUFSORT_HIP_LIB; else export MSUFSORT_HIP_LIB="$PWD/$lib"; fi
    for w in text dna; do
        echo "== $(basename $lib) $w: $(python tools/gpu_one.py $w 1073741823 0 4 2>/dev/null | grep -E 'build|errors' | tr '\n' ' ')"
    done
done
