#!/bin/bash
# round 6: k_chain_resolve loads the bulk of a segment's rows only when the first rows show a step
ulimit -c 0
O=gpurun_out/r6i; mkdir -p $O
( time timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_gpu_fuzz.py -q -m gpu -k "tandem or progression or deep or rank_array or fuzz or full_size" ) > $O/pytest_tandem.log 2>&1; tail -3 $O/pytest_tandem.log; grep FAILED $O/pytest_tandem.log | head
timeout 300 python tools/gpu_one.py dna_tandem 268435456 0 3 2>&1 | grep -E "build|errors"
CFG5_CHECK=1 timeout 900 python tools/gpu_cfg5.py 33 2 32 2>&1 | grep -E "build|errors" | cut -c1-300
