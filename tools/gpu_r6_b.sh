#!/bin/bash
# round 6: the several-segments-per-wave class-A sort (k_sort_mid_tiles): parity first, then timings against the one-segment instance
ulimit -c 0
O=gpurun_out/r6b; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
( time timeout 1500 python -m pytest tests/test_gpu_full.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
run() { # tag, env...
  tag=$1; shift
  for w in "text 1073741823" "dna 1073741823" "dna_tandem 268435456"; do
    set -- $w
    echo "== $tag $1 $2" >> $O/timings.txt
    env "${EXTRA[@]}" timeout 300 python tools/gpu_one.py $1 $2 0 3 >> $O/timings.txt 2>&1
  done
}
EXTRA=(A=1); run tiles_w4
EXTRA=(MSUFSORT_HIP_LIB=$PWD/msufsort_amd/lib/libmsufsort_hip_var_w5.so); run tiles_w5
EXTRA=(MSUFSORT_HIP_MID_SINGLE=1); run single
cat $O/timings.txt
MSUFSORT_HIP_LIB=$PWD/msufsort_amd/lib/libmsufsort_hip_prof_mid.so timeout 300 python tools/gpu_one.py text 1073741823 0 1 > $O/mid_prof_text_tiles.txt 2>&1
MSUFSORT_HIP_MID_SINGLE=1 MSUFSORT_HIP_LIB=$PWD/msufsort_amd/lib/libmsufsort_hip_prof_mid.so timeout 300 python tools/gpu_one.py text 1073741823 0 1 > $O/mid_prof_text_single.txt 2>&1
grep "class A" $O/mid_prof_text_tiles.txt | head -12; grep "class A" $O/mid_prof_text_single.txt | head -12
