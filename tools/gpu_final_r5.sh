#!/bin/bash
# End-of-round snapshot (round 5).  Everything lands in gpurun_out/final/ (tools/collect_r5.py copies what is to be judged into
# profiles/r05_* and rebuilds profiles/pmc_traffic.json for THIS build of the library).  "quick": without the full pytest run.
ulimit -c 0
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
if [ "$1" != "quick" ]; then
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -18 $O/pytest_gpu.log
fi
( time timeout 900 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_time.txt | grep -v amdgpu.ids | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json; tail -3 $O/bench_time.txt
timeout 300 python bench.py --steps 20 --warmup 10 --size 268435456 --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_256MiB.json
MSUFSORT_HIP_TWO_STAGE=-1 timeout 300 python bench.py --steps 3 --warmup 1 --workload text --op sa,bwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_sort_all.json
timeout 300 python bench.py --steps 3 --warmup 1 --workload dna --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna.json
timeout 300 python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --op sa,bwt,ibwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna_tandem_256MiB.json
export MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1
timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --size 268435456 --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_256MiB.json; cut -c1-200 $O/bench_2ranks_one_gpu_256MiB.json
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --size 268435456 --workload text --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_text_256MiB_two_stage_sharded.json
timeout 600 python bench.py --gpus 4 --steps 1 --warmup 0 --size 16777216 --workload dna_tandem --index int64 --no-cpu --check-reference 2>/dev/null | tail -1 > $O/bench_4ranks_one_gpu_int64_dna_tandem_16MiB.json; cut -c1-200 $O/bench_4ranks_one_gpu_int64_dna_tandem_16MiB.json
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --size 268435456 --op sa,fbwt --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_256MiB_sa_fbwt.json; cut -c1-200 $O/bench_2ranks_one_gpu_256MiB_sa_fbwt.json
unset MSUFSORT_BENCH_BACKEND MSUFSORT_BENCH_ONE_DEVICE
timeout 600 python tools/gpu_r4_sizes.py 2>&1 | grep "MiB:" > $O/sizes.txt; tail -4 $O/sizes.txt
tools/gpu_prof_bench.sh kernel_stats_random --steps 3 --warmup 1 --no-configs --no-host > $O/kernel_stats_random.txt 2>&1; cp gpurun_out/prof/kernel_stats_random.csv $O/
tools/gpu_prof_bench.sh kernel_stats_text --workload text --op sa,fbwt,ibwt,lcp --steps 2 --warmup 1 --no-host > $O/kernel_stats_text.txt 2>&1; cp gpurun_out/prof/kernel_stats_text.csv $O/
tools/gpu_prof_bench.sh kernel_stats_dna --workload dna --steps 2 --warmup 1 --no-host > $O/kernel_stats_dna.txt 2>&1; cp gpurun_out/prof/kernel_stats_dna.csv $O/
timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random.txt --no-configs --no-host > /dev/null 2>&1; head -6 $O/pmc_traffic_random.txt
timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_sa.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1; head -8 $O/pmc_traffic_text_sa.txt
MSUFSORT_HIP_KEY1=-1 timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_sa_key1_off.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_ibwt_lcp.txt --workload text --op sa,bwt,ibwt,lcp --no-configs --no-host > /dev/null 2>&1
timeout 400 bash tools/gpu_pmc_sq.sh $O/pmc_sq_text.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1
timeout 400 bash tools/gpu_pmc_sq.sh $O/pmc_sq_random.txt --no-configs --no-host > /dev/null 2>&1
timeout 300 python tools/gpu_key1.py 2>&1 | grep -E "RESULT" > $O/key1_ab.txt; cat $O/key1_ab.txt
timeout 200 python tools/gpu_verbose_any.py text 1073741823 2>&1 | grep -E "msufsort_hip|errors" > $O/text_rounds.txt
timeout 200 python tools/gpu_verbose_any.py dna_tandem 268435456 -1 2>&1 | grep -E "msufsort_hip|errors" > $O/tandem_rounds.txt
bash tools/gpu_trace_py.sh trace_text 700 tools/gpu_one.py text 1073741823 0 2 > /dev/null 2>&1; grep -v "k_zero_idx\|k_tiles\|k_segscan\|fillBuffer\|copyBuffer\|k_copy_idx\|k_ind_fused" gpurun_out/prof/trace_text.txt > $O/trace_text_two_stage.txt
cp gpurun_out/r5_two_stage_sweep.txt $O/two_stage_sweep.txt 2>/dev/null
ls -la $O | head -70
