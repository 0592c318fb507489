#!/bin/bash
ulimit -c 0
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/gpu_prof_any.sh text 268435456 2>&1 | grep -E "SA:|RESULT|sort_mid|sort_tiny|refill|isa_init|k_children|k_carry|sort_fast"
