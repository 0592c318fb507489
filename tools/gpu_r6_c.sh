#!/bin/bash
# round 6: the characters in front of the B* suffixes picked up by the gathering sorts (GatherSpec::pc_out): parity, then A/B timings
ulimit -c 0
O=gpurun_out/r6c; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
( time timeout 1800 python -m pytest tests/test_gpu_full.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log; grep FAILED $O/pytest.log | head
run() { tag=$1; shift
  for w in "text 1073741823" "dna 1073741823"; do set -- $w
    echo "== $tag $1 $2" >> $O/timings.txt
    env "${EXTRA[@]}" timeout 300 python tools/gpu_one.py $1 $2 0 3 >> $O/timings.txt 2>&1
  done; }
EXTRA=(A=1); run pcw_on
EXTRA=(MSUFSORT_HIP_NO_PCW=1); run pcw_off
grep -E "^==|build [12]|errors" $O/timings.txt | paste - - - - | sed 's/\/opt[^ ]*//' | cut -c1-150
MSUFSORT_HIP_VERBOSE_ANY=1 timeout 300 python - > $O/text_verbose.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, msufsort_amd as M
from msufsort_amd import gen
n = (1 << 30) - 1
t = gen.text_bytes(n, 3)
d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); d[:n] = torch.from_numpy(t).cuda()
ctx = M.DeviceContext(0); sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
ctx.make_sa(d, n, sa); ctx.make_sa(d, n, sa, verbose=1)
tm = ctx.timings(); print("total", tm.total_ms, "induction", tm.other_ms, "front", tm.front_ms)
PY
grep -E "round|two-stage|total" $O/text_verbose.txt | cut -c1-230 | head -30
