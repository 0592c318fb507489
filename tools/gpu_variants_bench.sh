#!/bin/bash
# headline bench of alternative builds (build/lib_NAME.so): tools/gpu_variants_bench.sh NAME...
for v in "$@"; do
  export MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$v.so
  python bench.py --steps 10 --warmup 3 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['valid'], d['ms_per_step'], d['phases_ms'])"
done
