#!/bin/bash
# round 6: k_sort_bits clears its words while the rows go out (one barrier and one phase less per segment)
ulimit -c 0
O=gpurun_out/r6z; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full.py -x -q -m gpu ) > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1; grep FAILED $O/pytest.log | head
MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip_prof_mid.so timeout 300 python tools/gpu_one.py random 1073741823 0 2 2>&1 | grep -E "bits prof|build" | cut -c1-400 > $O/bits_prof.txt
cat $O/bits_prof.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-host 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'], d['valid'])"
for n in 268435456 536870912 805306368 1342177280 2147483646; do timeout 300 python tools/gpu_one.py random $n 0 3 2>&1 | grep -E "build 2|errors" | paste - - ; done
