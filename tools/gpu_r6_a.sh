#!/bin/bash
# round 6, first contact: the int64 one-rank RCCL test at n = 2^31 + 12,345 and the driver's bench command with cfg2 / cfg5 in the line
ulimit -c 0
O=gpurun_out/r6a; mkdir -p $O
python -c "from msufsort_amd import _lib; print(_lib.lib().msufsort_hip_build_id().decode())" > $O/build_id.txt 2>/dev/null; cat $O/build_id.txt
( time timeout 3000 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -k "needs_them" ) > $O/test_int64_rccl.log 2>&1; tail -5 $O/test_int64_rccl.log
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6a/bench.json").read().strip().splitlines()[-1])
    print("headline", d["ms_per_step"], d["value"], d["valid"], d["cpu_baseline"].get("value"), d["cpu_baseline"].get("runs_at_reported_threads"), d["cpu_baseline"].get("single_thread"))
    for k, v in d["configs"].items():
        print(k, {kk: v[kk] for kk in v if kk in ("valid", "sa_ms", "sa_MBps", "skipped", "error", "doubling_ms", "doubling_steps", "first_build_ms_with_allocations", "generate_and_upload_s", "check_s", "phases_ms", "inverse_bwt_ms", "forward_bwt_ms")})
        if "roofline" in v: print("   roofline", v["roofline"])
        if "cpu_baseline" in v: print("   cpu", {kk: v["cpu_baseline"].get(kk) for kk in ("value", "cores", "error")})
except Exception as e:
    print("no line:", e)
PY
