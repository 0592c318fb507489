#!/bin/bash
# round 6: full GPU suite + the driver's bench command on the current tree
ulimit -c 0
O=gpurun_out/r6f; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
( time timeout 2400 python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log; grep FAILED $O/pytest_gpu.log | head
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6f/bench.json").read().strip().splitlines()[-1])
    print("headline", d["ms_per_step"], d["value"], d["valid"], {k: v["ms"] for k, v in d["kernels"].items()})
    for k, v in d["configs"].items():
        print(k, {kk: v[kk] for kk in v if kk in ("valid", "sa_ms", "sa_MBps", "skipped", "error", "doubling_ms", "phases_ms", "inverse_bwt_ms", "forward_bwt_ms", "lcp_ms")})
        if "two_stage" in v: print("   ", v["two_stage"])
        if "request_bound" in v: print("   ", {a: b.get("frac_of_measured_ceiling") for a, b in v["request_bound"].items()})
        if "roofline" in v: print("   roofline", v["roofline"]["kernel"][:50], v["roofline"]["frac"], v["roofline"]["launch_ms"])
except Exception as e:
    print("no line:", e)
PY
./tools/microbench/bin/exp_lds_hist_ceiling 2>&1 | tail -1
