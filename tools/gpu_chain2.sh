#!/bin/bash
ulimit -c 0
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_full.py -x -q -m gpu -k "dna" 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --no-cpu 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tandem', d['ms_per_step'], d['valid'], {k: round(v,2) for k,v in d['phases_ms'].items()})"
MSUFSORT_TEST_VERBOSE=0 timeout 1300 python -m pytest tests/test_gpu_big.py -q -m gpu -s 2>&1 | grep -E "config 5|n=|passed|failed|Error|skipped" | cut -c1-400
