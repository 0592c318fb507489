#!/bin/bash
# Compare alternative builds of the library (make -C msufsort_amd/csrc variant VAR_FLAGS=... VAR_TAG=...) on the headline and on
# the sizes / workloads that share its kernels: parity subset first, then phase times.
#   tools/gpu_variant_bench.sh <lib> [<lib> ...]      ("default" = the product library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
for lib in "$@"; do
    tag=$(basename "$lib" .so)
    if [ "$lib" = default ]; then unset MSUFSORT_HIP_LIB; tag=default; else export MSUFSORT_HIP_LIB="$PWD/$lib"; fi
    echo "=== $tag"
    timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu -k "fuzz_kind or radix17 or two_stage or golden or small or dna or text" 2>&1 | tail -3
    for args in "" "--workload text" "--workload dna" "--size 2147483646" "--size 268435456"; do
        timeout 600 python bench.py --no-host --no-cpu --no-configs --steps 5 --warmup 2 $args 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'): continue
    j = json.loads(line)
    ph = j.get('phases_ms') or j.get('config', {}).get('phases_ms') or {}
    print('  ', '$args' or 'random 1 GiB', 'ms/step', j.get('ms_per_step'), {k: ph[k] for k in ph if k.startswith(('k_hist16', 'k_scatter0', 'k_partition', 'bucket', 'key rounds', 'induction', 'device_total'))}, 'valid', j.get('valid'))
"
    done
done 2>&1 | tee gpurun_out/variants/variant_bench.txt
