#!/bin/bash
# inverse-BWT timing of alternative builds (build/lib_NAME.so): tools/gpu_variants_ibwt.sh NAME...
for v in "$@"; do
  export MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$v.so
  python bench.py --workload text --op sa,bwt,ibwt --steps 2 --warmup 1 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['valid'], d['ops_ms'], d['ibwt'])"
done
