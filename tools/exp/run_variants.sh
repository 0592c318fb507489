#!/bin/bash
# usage: run_variants.sh NAME... : benchmarks msufsort_amd/lib built as tools/exp/bin/lib_NAME.so (timing only)
ulimit -c 0
cp msufsort_amd/lib/libmsufsort_hip.so /tmp/lib_base.so
for n in base "$@"; do
  if [ $n = base ]; then cp /tmp/lib_base.so msufsort_amd/lib/libmsufsort_hip.so; else cp tools/exp/bin/lib_$n.so msufsort_amd/lib/libmsufsort_hip.so; fi
  echo "== $n"
  timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu ${BENCH_ARGS:-} 2>&1 | grep -v amdgpu.ids | tee /tmp/bench_raw.txt | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms', d['phases_ms'], 'valid', d.get('valid'))"
  grep "prof\]" /tmp/bench_raw.txt | tail -2
done
cp /tmp/lib_base.so msufsort_amd/lib/libmsufsort_hip.so
