#!/bin/bash
# Second snapshot of round 5, after k_hist16's DENSE mode: what depends on the library build (bench line, kernel stats, PMC passes,
# key-1 A/B, verbose rounds) again, merged over gpurun_out/final/ of tools/gpu_final_r5.sh.
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_time.txt | grep -v amdgpu.ids | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -3 $O/bench_time.txt
timeout 300 python bench.py --steps 3 --warmup 1 --workload dna --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna.json
timeout 300 python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --op sa,bwt,ibwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna_tandem_256MiB.json
MSUFSORT_HIP_TWO_STAGE=-1 timeout 300 python bench.py --steps 3 --warmup 1 --workload text --op sa,bwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_sort_all.json
tools/gpu_prof_bench.sh kernel_stats_random --steps 3 --warmup 1 --no-configs --no-host > $O/kernel_stats_random.txt 2>&1; cp gpurun_out/prof/kernel_stats_random.csv $O/
tools/gpu_prof_bench.sh kernel_stats_text --workload text --op sa,fbwt,ibwt,lcp --steps 2 --warmup 1 --no-host > $O/kernel_stats_text.txt 2>&1; cp gpurun_out/prof/kernel_stats_text.csv $O/
tools/gpu_prof_bench.sh kernel_stats_dna --workload dna --steps 2 --warmup 1 --no-host > $O/kernel_stats_dna.txt 2>&1; cp gpurun_out/prof/kernel_stats_dna.csv $O/
bash tools/gpu_rekey_r5.sh
timeout 300 python tools/gpu_key1.py 2>&1 | grep -E "RESULT" > $O/key1_ab.txt; cat $O/key1_ab.txt
timeout 200 python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -E "msufsort_hip|errors" > $O/text_rounds.txt
bash tools/gpu_trace_py.sh trace_text 700 tools/gpu_one.py text 1073741823 0 2 > /dev/null 2>&1; grep -v "k_zero_idx\|k_tiles\|k_segscan\|fillBuffer\|copyBuffer\|k_copy_idx\|k_ind_fused" gpurun_out/prof/trace_text.txt > $O/trace_text_two_stage.txt
timeout 400 bash tools/gpu_pmc_sq.sh $O/pmc_sq_text.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1
ls -la $O | head -60
