#!/bin/bash
# End-of-round snapshot (round 2): parity tests, bench lines (headline with CPU baseline, 256 MiB, text sa+bwt+ibwt, DNA,
# tandem DNA), kernel-trace stats, PMC traffic, the > 2^32 test, host-pointer rates, micro-benchmarks.  Everything lands in
# gpurun_out/final/ (copy what is to be judged into profiles/).
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
python bench.py --steps 20 --warmup 10 --size 268435456 --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_256MiB.json
python bench.py --steps 3 --warmup 1 --workload text --op sa,bwt,ibwt,lcp --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_cfg3_cfg4.json; cut -c1-200 $O/bench_text_cfg3_cfg4.json
MSUFSORT_HIP_TWO_STAGE=-1 python bench.py --steps 3 --warmup 1 --workload text --op sa,bwt --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_sort_all.json
python bench.py --steps 5 --warmup 2 --workload text --op sa,fbwt,ibwt --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_forward_bwt.json
python tools/gpu_two_stage_sweep.py text 4 16 32 64 128 256 1023 > $O/two_stage_sweep.txt 2>&1; python tools/gpu_two_stage_sweep.py dna 64 256 1023 >> $O/two_stage_sweep.txt 2>&1; python tools/gpu_two_stage_sweep.py dna_tandem 256 >> $O/two_stage_sweep.txt 2>&1; grep MiB $O/two_stage_sweep.txt
python bench.py --steps 3 --warmup 1 --workload dna --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna.json
python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --op sa,bwt,ibwt --no-cpu 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna_tandem_256MiB.json
tools/gpu_prof_bench.sh kernel_stats_random --steps 3 --warmup 1 > $O/kernel_stats_random.txt 2>&1; cp gpurun_out/prof/kernel_stats_random.csv $O/
tools/gpu_prof_bench.sh kernel_stats_text --workload text --op sa,bwt,ibwt --steps 2 --warmup 1 > $O/kernel_stats_text.txt 2>&1; cp gpurun_out/prof/kernel_stats_text.csv $O/
MSUFSORT_HIP_TWO_STAGE=-1 tools/gpu_prof_bench.sh kernel_stats_text_sort_all --workload text --op sa --steps 2 --warmup 1 > $O/kernel_stats_text_sort_all.txt 2>&1; cp gpurun_out/prof/kernel_stats_text_sort_all.csv $O/
tools/gpu_prof_bench.sh kernel_stats_tandem --workload dna_tandem --size 268435456 --steps 2 --warmup 1 > $O/kernel_stats_tandem.txt 2>&1; cp gpurun_out/prof/kernel_stats_tandem.csv $O/
bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random.txt > /dev/null 2>&1; cat $O/pmc_traffic_random.txt
bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_ibwt.txt --workload text --op sa,bwt,ibwt > /dev/null 2>&1
MSUFSORT_TEST_VERBOSE=0 timeout 900 python -m pytest tests/test_gpu_big.py -q -m gpu -s > $O/big_2pow32.log 2>&1; grep "n=" $O/big_2pow32.log
python tools/gpu_hostapi.py > $O/hostapi.txt 2>&1; python tools/gpu_hostapi.py 1073741823 text >> $O/hostapi.txt 2>&1; tail -12 $O/hostapi.txt
python tools/gpu_hist.py > $O/hist16_shapes.txt 2>&1
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/exp_lds_order.hip -o /tmp/exp_lds_order 2>/dev/null && timeout 100 /tmp/exp_lds_order ) > $O/microbench_lds_order.txt 2>&1
