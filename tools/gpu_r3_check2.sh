#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage or generated_golden or bucket or large_random" 2>&1 | tail -3
for w in dna dna_tandem; do
  sz=1073741823; ops=sa; [ $w = dna_tandem ] && sz=268435456 && ops=sa,bwt,ibwt
  python bench.py --steps 3 --warmup 1 --workload $w --size $sz --op $ops --no-cpu 2>gpurun_out/b.err | tail -1 > gpurun_out/bench_$w.json
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_$w.json')); print('$w', d['ms_per_step'], d.get('valid'), {k: round(v,2) for k,v in d['phases_ms'].items()}, d.get('ops_ms'), d.get('fallbacks'))" || tail -3 gpurun_out/b.err
done
python bench.py --steps 3 --warmup 1 --workload text --op sa --no-cpu 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('text', d['ms_per_step'], {k: round(v,2) for k,v in d['phases_ms'].items()})"
