#!/bin/bash
ulimit -c 0
for l in "$@"; do echo "== $l"; bash tools/gpu_exp.sh $l 2>&1 | grep -E "sort_fast|valid" ; done
