#!/bin/bash
# round 6: ring chunks copied in 4 MiB parts (first H2D chunk / last D2H chunk shared by the workers): bench host legs A/B
ulimit -c 0
O=gpurun_out/r6q; mkdir -p $O
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu --no-cfg5 > $O/bench_parts.json 2> $O/bench_parts.err
MSUFSORT_HIP_RING_PART=64 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu --no-cfg5 > $O/bench_whole.json 2> $O/bench_whole.err
python3 - <<'PY'
import json
for f in ("parts", "whole"):
    d = json.loads([l for l in open(f"gpurun_out/r6q/bench_{f}.json") if l.startswith("{")][-1])
    for nm, e in (("random", d["end_to_end_host"]), ("text", d["configs"]["end_to_end_host"])):
        print(f, nm, "c_abi", e["c_abi"].get("sa_ms"), e["c_abi"].get("valid"), "header sa", e["cpp_header"].get("sa_ms"), "fbwt", e["cpp_header"].get("forward_bwt_ms"), "ibwt", e["cpp_header"].get("inverse_bwt_ms"), e["cpp_header"].get("valid"), "floor", e["pcie_floor"]["sa_floor_ms"])
    print(f, "headline", d["ms_per_step"], "cfg3", d["configs"]["cfg3"]["sa_ms"], d["configs"]["cfg3"]["forward_bwt_ms"], d["configs"]["cfg3"]["valid"])
PY
