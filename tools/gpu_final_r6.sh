#!/bin/bash
# Round 6, final evidence on ONE library build: full GPU suite, the driver's bench command, kernel stats, PMC traffic (re-keyed to the build),
# SQ counters, phase clocks, the micro-benchmark behind k_hist16's ceiling, small multi-rank lines.  Results: gpurun_out/final6/ -> tools/collect_r6.py.
ulimit -c 0
O=gpurun_out/final6; mkdir -p $O
R=$GRAFT_REPO_ROOT
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
stats() { # name, divisor, program args...
  name=$1; div=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- "$@" > /tmp/p.log 2>&1; grep -E "^build|errors|generated" /tmp/p.log | cut -c1-300; find /tmp/praw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/$O/$name.csv )
  python3 - $O/$name.csv $div "$*" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1]))); div = float(sys.argv[2])
tot = sum(float(r[2]) for r in rows[1:])
print("rocprofv3 --kernel-trace --stats --", sys.argv[3])
print("all kernels %.3f ms per build (%g builds in the process; incl. the checker's / generator's kernels)" % (tot / 1e6 / div, div))
for r in rows[1:30]:
    print("  %10.3f ms/build %9.1f launches  avg %9.1f us  %s" % (float(r[2]) / 1e6 / div, float(r[1]) / div, float(r[3]) / 1e3, r[0][:130]))
PY
}
if [ "${SKIP_SUITE:-0}" != 1 ]; then ( time timeout 2400 python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; grep FAILED $O/pytest_gpu.log | head; fi
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) 2> $O/bench_time.txt | grep -v amdgpu.ids | tail -1 > $O/bench.json; cut -c1-160 $O/bench.json; tail -3 $O/bench_time.txt
timeout 300 python bench.py --steps 3 --warmup 1 --workload dna --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna.json
timeout 300 python bench.py --steps 3 --warmup 1 --workload dna_tandem --size 268435456 --op sa,bwt,ibwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_dna_tandem_256MiB.json
MSUFSORT_HIP_TWO_STAGE=-1 timeout 300 python bench.py --steps 3 --warmup 1 --workload text --op sa,bwt --no-cpu --no-host 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_text_sort_all.json
# kernel stats
stats kernel_stats_random 4 python3 $R/bench.py --steps 3 --warmup 1 --no-configs --no-host --no-cpu > $O/kernel_stats_random.txt 2>&1
stats kernel_stats_text 3 python3 $R/bench.py --workload text --op sa,fbwt,ibwt,lcp --steps 2 --warmup 1 --no-host --no-cpu > $O/kernel_stats_text.txt 2>&1
stats kernel_stats_dna 3 python3 $R/bench.py --workload dna --steps 2 --warmup 1 --no-host --no-cpu > $O/kernel_stats_dna.txt 2>&1
stats kernel_stats_dna_tandem_256MiB 4 python3 $R/tools/gpu_one.py dna_tandem 268435456 0 4 > $O/kernel_stats_dna_tandem_256MiB.txt 2>&1
stats kernel_stats_cfg5 2 python3 $R/tools/gpu_cfg5.py 33 2 32 > $O/kernel_stats_cfg5.txt 2>&1
head -12 $O/kernel_stats_random.txt | cut -c1-150
# PMC traffic: FETCH_SIZE / WRITE_SIZE in separate passes, ONE build each
timeout 600 bash tools/gpu_pmc_cmd.sh $O/pmc_traffic_random.txt python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-configs --no-host > /dev/null 2>&1; head -5 $O/pmc_traffic_random.txt
timeout 600 bash tools/gpu_pmc_cmd.sh $O/pmc_traffic_random_256MiB.txt python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-configs --no-host --size 268435456 > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_cmd.sh $O/pmc_traffic_text_sa.txt python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --workload text --op sa --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_cmd.sh $O/pmc_traffic_dna.txt python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --workload dna --op sa --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_cmd.sh $O/pmc_traffic_text_ibwt_lcp.txt python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --workload text --op sa,bwt,ibwt,lcp --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_pmc_cmd.sh $O/pmc_traffic_dna_tandem_256MiB.txt python3 $R/tools/gpu_one.py dna_tandem 268435456 0 1 > /dev/null 2>&1
CFG5_CHECK=0 PHASE_SPLIT="k_isa_from_slice<true>" timeout 1500 bash tools/gpu_pmc_cmd.sh $O/pmc_traffic_cfg5.txt python3 $R/tools/gpu_cfg5.py 33 1 32 > /dev/null 2>&1; grep PHASE $O/pmc_traffic_cfg5.txt
# SQ counters
timeout 600 bash tools/gpu_sq_cmd.sh $O/pmc_sq_text.txt python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --workload text --op sa --no-configs --no-host > /dev/null 2>&1
timeout 600 bash tools/gpu_sq_cmd.sh $O/pmc_sq_dna_tandem_256MiB.txt python3 $R/tools/gpu_one.py dna_tandem 268435456 0 1 > /dev/null 2>&1
timeout 600 bash tools/gpu_sq_cmd.sh $O/pmc_sq_random.txt python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-configs --no-host > /dev/null 2>&1
# phase clocks of the LDS sorts (text), tiles against the one-segment instance
MSUFSORT_HIP_LIB=$R/msufsort_amd/lib/libmsufsort_hip_prof_mid.so timeout 300 python tools/gpu_one.py text 1073741823 0 1 2>&1 | grep "mid prof" > $O/mid_prof_text.txt
MSUFSORT_HIP_MID_SINGLE=1 MSUFSORT_HIP_LIB=$R/msufsort_amd/lib/libmsufsort_hip_prof_mid.so timeout 300 python tools/gpu_one.py text 1073741823 0 1 2>&1 | grep "mid prof" > $O/mid_prof_text_single.txt
# A/B of the round's levers on this build
for e in A=1 MSUFSORT_HIP_MID_SINGLE=1 MSUFSORT_HIP_NO_PCW=1 MSUFSORT_HIP_NO_TINY2=1 MSUFSORT_HIP_IND_PC_RAW=1; do for w in "text 1073741823" "dna 1073741823" "dna_tandem 268435456"; do set -- $w
  echo "== $e $1" >> $O/levers_ab.txt; env $e timeout 300 python tools/gpu_one.py $1 $2 0 3 2>&1 | grep -E "build [12]|errors" >> $O/levers_ab.txt; done; done
paste - - - - < $O/levers_ab.txt | cut -c1-120
MSUFSORT_TEST_VERBOSE=1 timeout 300 python - > $O/text_rounds.txt 2>&1 <<'PY'
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
# host-pointer calls on the 1 GiB text: the library's own timeline (results leave region by region), then the same calls with the round's host-side levers off
MSUFSORT_HIP_HOST_TRACE=1 timeout 600 python tools/gpu_host_text.py text 1073741823 2 sa,fbwt 2>&1 | grep -v -E "slice copy|amdgpu.ids" > $O/host_trace_text.txt
MSUFSORT_HIP_NO_EARLY_B=1 MSUFSORT_HIP_NO_BWT_RIDE=1 MSUFSORT_HIP_RING_PART=64 timeout 600 python tools/gpu_host_text.py text 1073741823 2 sa,fbwt 2>&1 | grep -v -E "host trace|amdgpu.ids" > $O/host_text_levers_off.txt
grep -E "rep [12]" $O/host_trace_text.txt $O/host_text_levers_off.txt | cut -c1-160
for e in A=1 MSUFSORT_HIP_NO_BWT_RIDE=1; do echo "== $e" >> $O/fbwt_dev_ab.txt; for w in text dna; do env $e timeout 300 python tools/gpu_fbwt_dev.py $w 1073741823 4 2>&1 | grep forward_bwt_dev >> $O/fbwt_dev_ab.txt; done; done; cat $O/fbwt_dev_ab.txt
./tools/microbench/bin/exp_lds_hist_ceiling > $O/microbench_lds_hist_ceiling.txt 2>&1; cat $O/microbench_lds_hist_ceiling.txt
# small multi-rank lines (gloo, all ranks on the one GPU): plumbing, NOT performance figures
export MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1
timeout 400 python bench.py --gpus 2 --steps 1 --warmup 0 --size 67108864 --no-cpu 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_64MiB_sub_shards.json; cut -c1-200 $O/bench_2ranks_one_gpu_64MiB_sub_shards.json
MSUFSORT_DIST_SUBSHARDS=2 timeout 400 python bench.py --gpus 4 --steps 1 --warmup 0 --size 16777216 --workload dna_tandem --index int64 --op sa,fbwt --no-cpu --check-reference 2>/dev/null | tail -1 > $O/bench_4ranks_one_gpu_int64_dna_tandem_16MiB.json
unset MSUFSORT_BENCH_BACKEND MSUFSORT_BENCH_ONE_DEVICE
ls -la $O | head -70
# robustness on the same build
( timeout 900 python tools/gpu_determinism.py ) 2>&1 | grep -v amdgpu > $O/determinism.txt; tail -3 $O/determinism.txt
( timeout 900 python tools/gpu_stress.py ) 2>&1 | grep -v amdgpu > $O/stress.txt; tail -2 $O/stress.txt
( timeout 900 python tools/gpu_patho.py ) 2>&1 | grep -v amdgpu > $O/patho.txt; tail -2 $O/patho.txt
( timeout 900 python tools/gpu_real_corpus.py ) 2>&1 | grep -v amdgpu > $O/real_corpus.txt; tail -1 $O/real_corpus.txt
( timeout 1200 python tools/gpu_stress_big.py ) 2>&1 | grep -v amdgpu > $O/stress_big.txt; tail -2 $O/stress_big.txt
ls $O | wc -l
