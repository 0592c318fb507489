#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "two_stage" 2>&1 | tail -5
python -m pytest tests/test_gpu_dist.py -x -q -k "two_stage_sharded or two_ranks_one_gpu" 2>&1 | tail -5
python -m pytest tests/test_gpu_big.py -x -q -k "class_limits or hist17" 2>&1 | tail -3
