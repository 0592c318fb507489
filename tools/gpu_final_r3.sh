#!/bin/bash
# End-of-round snapshot (round 3): parity tests, the driver's bench line (headline + configs 3/4 + CPU legs), 256 MiB, DNA,
# tandem DNA, 2 ranks on one GPU, kernel-trace stats, PMC traffic + SQ counters, micro-benchmarks.  Everything lands in
# gpurun_out/final/ (copy what is to be judged into profiles/).
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
MSUFSORT_TEST_VERBOSE=0 python -m pytest tests/test_gpu_big.py -q -m gpu -s 2>&1 | grep -E "config 5|n=|passed|failed" > $O/big_inputs.log; cat $O/big_inputs.log
( time python bench.py --steps 20 --warmup 5 ) 2> $O/bench_time.txt | grep -v amdgpu.ids | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json; tail -3 $O/bench_time.txt
python bench.py --steps 20 --warmup 10 --size 268435456 --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_256MiB.json
MSUFSORT_HIP_BUCKET_SORT=fast2 python bench.py --steps 10 --warmup 3 --no-cpu --no-configs 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_fast2.json
MSUFSORT_HIP_TWO_STAGE=-1 python bench.py --steps 3 --warmup 1 --workload text --op sa,bwt --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_sort_all.json
python bench.py --steps 3 --warmup 1 --workload dna --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna.json
python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --op sa,bwt,ibwt --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna_tandem_256MiB.json
MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --size 268435456 --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_256MiB.json; cut -c1-200 $O/bench_2ranks_one_gpu_256MiB.json
tools/gpu_prof_bench.sh kernel_stats_random --steps 3 --warmup 1 --no-configs > $O/kernel_stats_random.txt 2>&1; cp gpurun_out/prof/kernel_stats_random.csv $O/
tools/gpu_prof_bench.sh kernel_stats_text --workload text --op sa,fbwt,ibwt,lcp --steps 2 --warmup 1 > $O/kernel_stats_text.txt 2>&1; cp gpurun_out/prof/kernel_stats_text.csv $O/
tools/gpu_prof_bench.sh kernel_stats_dna --workload dna --steps 2 --warmup 1 > $O/kernel_stats_dna.txt 2>&1; cp gpurun_out/prof/kernel_stats_dna.csv $O/
tools/gpu_prof_bench.sh kernel_stats_tandem --workload dna_tandem --size 268435456 --steps 2 --warmup 1 > $O/kernel_stats_tandem.txt 2>&1; cp gpurun_out/prof/kernel_stats_tandem.csv $O/
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random.txt --no-configs > /dev/null 2>&1; cat $O/pmc_traffic_random.txt
timeout 600 bash tools/gpu_pmc_sq.sh $O/pmc_sq_random.txt --no-configs > /dev/null 2>&1; cat $O/pmc_sq_random.txt
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text.txt --workload text --op sa,fbwt,ibwt --no-configs > /dev/null 2>&1
python tools/gpu_verbose_any.py dna_tandem 268435456 -1 2>&1 | grep -E "msufsort_hip|errors" > $O/tandem_rounds.txt
for b in exp_lds_rates exp_lds_valu_overlap exp_bits_phases exp_bits_phases2 exp_random_lines exp_random_lines_k1; do timeout 120 ./tools/microbench/bin/$b > $O/microbench_$b.txt 2>&1; done
timeout 600 bash tools/gpu_pmc_sq.sh $O/pmc_sq_text.txt --workload text --op sa --no-configs > /dev/null 2>&1
timeout 600 bash tools/gpu_trace_py.sh trace_text_two_stage 700 tools/gpu_two_stage_only.py text 1073741823 2 > /dev/null 2>&1; cp gpurun_out/prof/trace_text_two_stage.txt $O/ 2>/dev/null
timeout 300 python tools/gpu_bits_sizes.py 2>&1 | grep "MiB:" > $O/bits_sizes.txt
MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip_prof.so python bench.py --steps 2 --warmup 1 --no-cpu --no-configs 2>&1 | grep "bits prof" | tail -1 > $O/bits_phase_clocks.txt; cat $O/bits_phase_clocks.txt
