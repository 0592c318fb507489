#!/bin/bash
# round 6: run statistics counted by k_types (k_maxrun's pass over the text gone): parity + timings
ulimit -c 0
O=gpurun_out/r6w; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "FAILED" $O/pytest.log | head
for w in "dna 1073741823 0" "text 1073741823 0"; do set -- $w
    echo "== $1 $2" >> $O/timings.txt
    timeout 300 python tools/gpu_one.py $1 $2 $3 4 2>&1 | grep -E "build [123]|errors" >> $O/timings.txt
    timeout 300 python tools/gpu_two_stage_only.py $1 $2 2>&1 | grep "two-stage" | cut -c1-200 >> $O/timings.txt
done
cat $O/timings.txt
