#!/bin/bash
# phase clocks of the induction's tiles (variant build -DIND_PROF, tools/experiments/ind_prof.diff) and the level launches at other grid sizes
ulimit -c 0
O=gpurun_out/r6indprof; mkdir -p $O; rm -f $O/ind_prof.txt
for g in 0 128 256 512 2048; do
for w in text dna; do
MSUFSORT_HIP_IND_GRID=$g MSUFSORT_HIP_LIB=$GRAFT_REPO_ROOT/msufsort_amd/lib/libmsufsort_hip_var_indprof.so timeout 300 python tools/gpu_two_stage_only.py $w 1073741823 2 2>&1 | grep -E "ind prof|induction ms" | tail -2 | sed "s/^/grid $g $w: /" >> $O/ind_prof.txt
MSUFSORT_HIP_IND_GRID=$g timeout 300 python tools/gpu_two_stage_only.py $w 1073741823 2 2>&1 | grep -E "induction ms" | tail -1 | sed "s/^/grid $g $w (product library): /" >> $O/ind_prof.txt
done; done
cat $O/ind_prof.txt
