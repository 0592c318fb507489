#!/bin/bash
# usage: run_any_variant.sh NAME cmd...  : runs cmd with msufsort_amd/lib replaced by tools/exp/bin/lib_NAME.so
ulimit -c 0
cp msufsort_amd/lib/libmsufsort_hip.so /tmp/lib_base.so
n=$1; shift
cp tools/exp/bin/lib_$n.so msufsort_amd/lib/libmsufsort_hip.so
"$@"
cp /tmp/lib_base.so msufsort_amd/lib/libmsufsort_hip.so
