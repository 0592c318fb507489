#!/bin/bash
# round 6: per-kernel time of the text build with the preceding characters carried by the sorts (on / off); the multi-rank flows with sub-shards
ulimit -c 0
O=gpurun_out/r6d; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
prof() { # name, env
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/praw && env "${EXTRA[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/praw -- python3 $GRAFT_REPO_ROOT/tools/gpu_one.py text 1073741823 0 3 > /tmp/p.log 2>&1; tail -3 /tmp/p.log; find /tmp/praw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/$O/$1.csv )
  python3 - $O/$1.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
tot = 0
for r in rows[1:]:
    tot += float(r[2])
print("all kernels %.2f ms per build (3 builds + checker)" % (tot / 1e6 / 3))
for r in rows[1:22]:
    print("  %9.3f ms/build %7.1f launches  %s" % (float(r[2]) / 1e6 / 3, float(r[1]) / 3, r[0][:110]))
PY
}
EXTRA=(A=1); prof kstats_text_pcw_on > $O/kstats_text_pcw_on.txt 2>&1
EXTRA=(MSUFSORT_HIP_NO_PCW=1); prof kstats_text_pcw_off > $O/kstats_text_pcw_off.txt 2>&1
head -24 $O/kstats_text_pcw_on.txt; head -24 $O/kstats_text_pcw_off.txt
( time timeout 1500 python -m pytest tests/test_gpu_dist.py -q -m gpu -k "not needs_them" ) > $O/pytest_dist.log 2>&1; tail -4 $O/pytest_dist.log; grep FAILED $O/pytest_dist.log | head
MSUFSORT_DIST_SUBSHARDS=3 MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --size 16777216 --no-cpu --check-reference > $O/bench_2ranks_sub3.json 2> $O/bench_2ranks_sub3.err; cut -c1-400 $O/bench_2ranks_sub3.json; tail -3 $O/bench_2ranks_sub3.err
MSUFSORT_DIST_SUBSHARDS=2 MSUFSORT_BENCH_BACKEND=gloo MSUFSORT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --size 4194304 --workload dna_tandem --index int64 --op sa,fbwt --no-cpu --check-reference > $O/bench_2ranks_sub2_tandem.json 2> $O/bench_2ranks_sub2_tandem.err; cut -c1-300 $O/bench_2ranks_sub2_tandem.json; tail -3 $O/bench_2ranks_sub2_tandem.err
