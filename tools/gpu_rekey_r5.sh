#!/bin/bash
# The PMC traffic passes alone for the present build of the library (then tools/collect_r5.py, which keys profiles/pmc_traffic.json
# to it), plus the small multi-rank lines.  Results: gpurun_out/final/ (merged over what tools/gpu_final_r5.sh left there).
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random.txt --no-configs --no-host > /dev/null 2>&1; head -6 $O/pmc_traffic_random.txt
timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_sa.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1
MSUFSORT_HIP_KEY1=-1 timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_sa_key1_off.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_ibwt_lcp.txt --workload text --op sa,bwt,ibwt,lcp --no-configs --no-host > /dev/null 2>&1
export MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1
timeout 400 python bench.py --gpus 2 --steps 1 --warmup 0 --size 67108864 --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_64MiB.json; cut -c1-200 $O/bench_2ranks_one_gpu_64MiB.json
ls -la $O | head -50
