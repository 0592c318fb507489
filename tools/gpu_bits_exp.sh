#!/bin/bash
# diagnostic builds of the bucket sort: phase clocks with / without record loads / row stores (results are then wrong on purpose)
for tag in "" 1 2 3; do
  echo "== BITS_EXP=$tag"
  MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip_prof$tag.so python bench.py --steps 1 --warmup 1 --no-cpu --no-configs 2>&1 | grep -E "bits prof|phases_ms" | tail -2 | sed -e 's/.*"phases_ms"/phases_ms/' | cut -c1-330
done
