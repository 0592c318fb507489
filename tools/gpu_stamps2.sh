#!/bin/bash
ulimit -c 0
cp msufsort_amd/lib/libmsufsort_hip.so /tmp/lib_backup.so
cp msufsort_amd/lib/libexp_stamps.so msufsort_amd/lib/libmsufsort_hip.so
python tools/gpu_configs.py text 268435456 2>&1 | grep -E "stamps|RESULT" | cut -c1-900
cp /tmp/lib_backup.so msufsort_amd/lib/libmsufsort_hip.so
