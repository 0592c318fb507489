#!/bin/bash
# round 6: parked deeper histograms (cfg5), sharded DNA parity
ulimit -c 0
O=gpurun_out/r6h; mkdir -p $O
( time timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -k "heavy or shard or sharded or logical or wide or sub_shards or four_and_eight" ) > $O/pytest_shards.log 2>&1; tail -4 $O/pytest_shards.log; grep FAILED $O/pytest_shards.log | head
CFG5_CHECK=1 timeout 900 python tools/gpu_cfg5.py 33 2 32 > $O/cfg5_plain.txt 2>&1; cat $O/cfg5_plain.txt | cut -c1-330
timeout 300 python tools/gpu_one.py dna_tandem 268435456 0 3 2>&1 | grep -E "build|errors"
timeout 300 python tools/gpu_one.py dna 1073741823 0 3 2>&1 | grep -E "build|errors"
