#!/bin/bash
# gr.sh <timeout-seconds> <command...>: gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged)
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > /tmp/gr_last.out 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then cat /tmp/gr_last.out; exit $rc; fi
  sleep 45
done
cat /tmp/gr_last.out; exit 3
