#!/bin/bash
mkdir -p gpurun_out
{
echo "--- torch alone"; python -c "import torch; print(torch.__version__, torch.version.hip, torch.cuda.is_available(), torch.cuda.device_count()); x=torch.zeros(4,device='cuda'); print(x.device)" 2>&1 | tail -3
echo "--- torch first then lib"; python -c "
import torch
x=torch.zeros(4,device='cuda'); print('torch ok', x.device)
import msufsort_amd as M
print('devices', M.device_count())
import numpy as np
print(M.make_suffix_array(b'banana'))
" 2>&1 | tail -5
echo "--- lib first then torch"; python -c "
import msufsort_amd as M
print('devices', M.device_count())
import torch
print(torch.cuda.is_available())
x=torch.zeros(4,device='cuda'); print('torch ok', x.device)
" 2>&1 | tail -5
echo "--- env"; env | grep -E "HIP|ROCR|HSA|CUDA|LD_LIBRARY" ; id; nproc; lscpu | grep -E "Model name|Socket|Thread|Core" ; free -g | head -2
} > gpurun_out/torch_probe.log 2>&1
cat gpurun_out/torch_probe.log
