#!/bin/bash
ulimit -c 0
cp msufsort_amd/lib/libmsufsort_hip.so /tmp/lib_backup.so
cp msufsort_amd/lib/libexp_stamps.so msufsort_amd/lib/libmsufsort_hip.so
python bench.py --steps 1 --warmup 0 --no-cpu 2>&1 | grep -E "stamps|valid" | cut -c1-400
cp /tmp/lib_backup.so msufsort_amd/lib/libmsufsort_hip.so
