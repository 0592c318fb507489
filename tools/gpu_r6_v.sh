#!/bin/bash
ulimit -c 0
O=gpurun_out/r6v; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense_numbers or two_stage" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "FAILED|Error|assert" $O/pytest.log | head
for w in "dna 1073741823 0" "text 1073741823 0"; do set -- $w
    echo "== $1 $2" >> $O/timings.txt
    timeout 300 python tools/gpu_one.py $1 $2 $3 4 2>&1 | grep -E "build [123]|errors" >> $O/timings.txt
done
paste - - - - - < $O/timings.txt | cut -c1-160
