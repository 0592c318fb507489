#!/bin/bash
# round 6: kernel stats of BASELINE config 5's workload (8 GiB tandem-repeat DNA, int64 rows, 32 logical shards on the one GPU) and of tandem DNA 2^28
ulimit -c 0
O=gpurun_out/r6e; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
stats() { # name, divisor, command...
  name=$1; div=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- "$@" > /tmp/p.log 2>&1; grep -E "build|errors|generated" /tmp/p.log; find /tmp/praw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/$name.csv )
  python3 - $O/$name.csv $div <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1]))); div = float(sys.argv[2])
tot = sum(float(r[2]) for r in rows[1:])
print("all kernels %.2f ms per build (%g builds; incl. the checker's and the generator's kernels)" % (tot / 1e6 / div, div))
for r in rows[1:28]:
    print("  %10.3f ms/build %9.1f launches  avg %9.1f us  %s" % (float(r[2]) / 1e6 / div, float(r[1]) / div, float(r[3]) / 1e3, r[0][:120]))
PY
}
stats kernel_stats_cfg5 2 python3 $GRAFT_REPO_ROOT/tools/gpu_cfg5.py 33 2 32 > $O/kernel_stats_cfg5.txt 2>&1; head -34 $O/kernel_stats_cfg5.txt
stats kernel_stats_dna_tandem_256MiB 4 python3 $GRAFT_REPO_ROOT/tools/gpu_one.py dna_tandem 268435456 0 4 > $O/kernel_stats_dna_tandem_256MiB.txt 2>&1; head -30 $O/kernel_stats_dna_tandem_256MiB.txt
# A/B of the preceding characters carried by the sorts (class A + tiny pool only)
for r in 1 2; do
for e in A=1 MSUFSORT_HIP_NO_PCW=1; do
  echo "== $e text" >> $O/pcw_ab.txt; env $e timeout 300 python tools/gpu_one.py text 1073741823 0 3 2>&1 | grep -E "build [12]|errors" >> $O/pcw_ab.txt
done; done
echo "== dna on" >> $O/pcw_ab.txt; timeout 300 python tools/gpu_one.py dna 1073741823 0 3 2>&1 | grep -E "build [12]|errors" >> $O/pcw_ab.txt
echo "== dna off" >> $O/pcw_ab.txt; MSUFSORT_HIP_NO_PCW=1 timeout 300 python tools/gpu_one.py dna 1073741823 0 3 2>&1 | grep -E "build [12]|errors" >> $O/pcw_ab.txt
cat $O/pcw_ab.txt | paste - - - - | cut -c1-120
./tools/microbench/bin/exp_lds_hist_ceiling > $O/microbench_lds_hist_ceiling.txt 2>&1; cat $O/microbench_lds_hist_ceiling.txt
