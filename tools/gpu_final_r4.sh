#!/bin/bash
# End-of-round snapshot (round 4): parity tests, the driver's bench line (headline + configs 3/4 + host-pointer legs + CPU legs),
# other workloads, rank flows on one GPU, kernel-trace stats, PMC traffic (random + text) and SQ counters, the size sweep, the
# host-path timeline and micro-benchmark.  Everything lands in gpurun_out/final/ (tools/collect_r4.py copies what is to be
# judged into profiles/r04_* and rebuilds profiles/pmc_traffic.json for THIS build of the library).
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
if [ "$1" != "quick" ]; then
python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -18 $O/pytest_gpu.log
fi
( time python bench.py --steps 20 --warmup 5 ) 2> $O/bench_time.txt | grep -v amdgpu.ids | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json; tail -3 $O/bench_time.txt
python bench.py --steps 20 --warmup 10 --size 268435456 --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_256MiB.json
MSUFSORT_HIP_TWO_STAGE=-1 python bench.py --steps 3 --warmup 1 --workload text --op sa,bwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_sort_all.json
python bench.py --steps 3 --warmup 1 --workload dna --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna.json
python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --op sa,bwt,ibwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna_tandem_256MiB.json
MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --size 268435456 --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_256MiB.json; cut -c1-200 $O/bench_2ranks_one_gpu_256MiB.json
MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 2 --warmup 1 --size 268435456 --workload text --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_text_256MiB_two_stage_sharded.json; cut -c1-200 $O/bench_2ranks_one_gpu_text_256MiB_two_stage_sharded.json
timeout 900 python tools/gpu_r4_sizes.py 2>&1 | grep "MiB:" > $O/sizes.txt; cat $O/sizes.txt
tools/gpu_prof_bench.sh kernel_stats_random --steps 3 --warmup 1 --no-configs --no-host > $O/kernel_stats_random.txt 2>&1; cp gpurun_out/prof/kernel_stats_random.csv $O/
tools/gpu_prof_bench.sh kernel_stats_text --workload text --op sa,fbwt,ibwt,lcp --steps 2 --warmup 1 --no-host > $O/kernel_stats_text.txt 2>&1; cp gpurun_out/prof/kernel_stats_text.csv $O/
tools/gpu_prof_bench.sh kernel_stats_2GiB --size 2147483646 --steps 2 --warmup 1 --no-host > $O/kernel_stats_2GiB.txt 2>&1; cp gpurun_out/prof/kernel_stats_2GiB.csv $O/
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random.txt --no-configs --no-host > /dev/null 2>&1; cat $O/pmc_traffic_random.txt
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random_2GiB.txt --size 2147483646 --no-configs --no-host > /dev/null 2>&1
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_sa.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1; cat $O/pmc_traffic_text_sa.txt
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_ibwt_lcp.txt --workload text --op sa,bwt,ibwt,lcp --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_sq.sh $O/pmc_sq_text.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_sq.sh $O/pmc_sq_random.txt --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_trace_fused.sh > $O/induction_level_durations.txt 2>&1
MSUFSORT_HIP_HOST_TRACE=1 timeout 300 python tools/gpu_host_fresh.py 1073741823 random 3 2>&1 | grep -v amdgpu.ids > $O/host_trace_random.txt
MSUFSORT_HIP_HOST_TRACE=1 timeout 300 python tools/gpu_host_fresh.py 1073741823 text 3 2>&1 | grep -v amdgpu.ids > $O/host_trace_text.txt
timeout 300 tools/microbench/bin/exp_host_xfer > $O/microbench_host_xfer.txt 2>&1
python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -E "msufsort_hip|errors" > $O/text_rounds.txt
python tools/gpu_verbose_any.py dna_tandem 268435456 -1 2>&1 | grep -E "msufsort_hip|errors" > $O/tandem_rounds.txt
timeout 900 python tools/gpu_stress.py 350000000 2>&1 | grep -v amdgpu.ids > $O/stress_350MB.txt; tail -14 $O/stress_350MB.txt
timeout 120 tools/microbench/bin/exp_h2d_fresh > $O/microbench_h2d_fresh.txt 2>&1
ls -la $O | head -70
