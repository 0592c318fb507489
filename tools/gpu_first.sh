#!/bin/bash
# first-contact run on the GPU box: each group in its own process with a timeout
mkdir -p gpurun_out
for g in hist lit sweep gen big; do
  echo "=== group $g" 
  timeout 300 python tools/gpu_probe.py $g 2>&1 | tail -80
  echo "exit: $?"
done > gpurun_out/probe.log 2>&1
tail -150 gpurun_out/probe.log
