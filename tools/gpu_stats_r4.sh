O=gpurun_out/final; mkdir -p $O
tools/gpu_prof_bench.sh kernel_stats_random --steps 3 --warmup 1 --no-configs --no-host > $O/kernel_stats_random.txt 2>&1; cp gpurun_out/prof/kernel_stats_random.csv $O/
tools/gpu_prof_bench.sh kernel_stats_text --workload text --op sa,fbwt,ibwt,lcp --steps 2 --warmup 1 --no-host > $O/kernel_stats_text.txt 2>&1; cp gpurun_out/prof/kernel_stats_text.csv $O/
tools/gpu_prof_bench.sh kernel_stats_2GiB --size 2147483646 --steps 2 --warmup 1 --no-host > $O/kernel_stats_2GiB.txt 2>&1; cp gpurun_out/prof/kernel_stats_2GiB.csv $O/
timeout 900 python tools/gpu_r4_sizes.py 2>&1 | grep "MiB:" > $O/sizes.txt; tail -3 $O/sizes.txt
grep -E "k_scatter0|k_partition<256>|k_sort_bits<1024" $O/kernel_stats_random.csv | cut -c1-60,200-330
