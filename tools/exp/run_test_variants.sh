#!/bin/bash
# usage: run_test_variants.sh "<pytest args>" NAME... : runs a pytest selection against msufsort_amd/lib built as tools/exp/bin/lib_NAME.so
ulimit -c 0
cp msufsort_amd/lib/libmsufsort_hip.so /tmp/lib_base.so
sel="$1"; shift
for n in base "$@"; do
  if [ $n = base ]; then cp /tmp/lib_base.so msufsort_amd/lib/libmsufsort_hip.so; else cp tools/exp/bin/lib_$n.so msufsort_amd/lib/libmsufsort_hip.so; fi
  echo "== $n"
  timeout 600 python -m pytest $sel 2>&1 | grep -v amdgpu | tail -4
done
cp /tmp/lib_base.so msufsort_amd/lib/libmsufsort_hip.so
