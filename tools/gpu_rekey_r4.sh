#!/bin/bash
# After a change that moves the library's build id without moving its kernels' behaviour (or after any small change, when the
# budget does not allow tools/gpu_final_r4.sh again): the driver's bench line and the PMC traffic passes for THIS build, so
# that profiles/pmc_traffic.json (tools/collect_r4.py) is keyed to the library that is shipped.  Results: gpurun_out/final/.
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random.txt --no-configs --no-host > /dev/null 2>&1; head -8 $O/pmc_traffic_random.txt
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_random_2GiB.txt --size 2147483646 --no-configs --no-host > /dev/null 2>&1
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_sa.txt --workload text --op sa --no-configs --no-host > /dev/null 2>&1
timeout 900 bash tools/gpu_pmc_traffic.sh $O/pmc_traffic_text_ibwt_lcp.txt --workload text --op sa,bwt,ibwt,lcp --no-configs --no-host > /dev/null 2>&1
ls -la $O
