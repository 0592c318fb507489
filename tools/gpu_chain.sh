#!/bin/bash
ulimit -c 0
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_full.py -x -q -m gpu -k "dna_tandem" 2>&1 | tail -2
python tools/gpu_verbose_any.py dna_tandem 268435456 -1 2>&1 | grep -E "progressions|errors|round 1[0-9] mode|round [3-9] mode" | cut -c1-200 | tail -16
python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --no-cpu 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tandem', d['ms_per_step'], d['valid'], {k: round(v,2) for k,v in d['phases_ms'].items()})"
