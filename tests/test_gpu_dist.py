"""GPU box (1 GPU): the full N>1 flow of bench.py - shard planning, per-rank sharded HIP sort, all-gatherv,
on-device validation - with 2, 4 and 8 ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device), and with ONE rank on
RCCL itself forced through the same code path (every process-group call of the flow, no point-to-point sends).  What has NOT run
anywhere yet: RCCL sends / receives and peer copies between two devices - no box this repo has met had a second GPU (the line
says what ran: `config.backend` / `config.rccl_ranks`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, MSUFSORT_BENCH_BACKEND="gloo", MSUFSORT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", MSUFSORT_DIST_SHARDED_HIST="1")
    # `python bench.py --gpus 2` DIRECTLY, the way the driver calls it: bench.py starts its two ranks itself
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--size", str(1 << 24), "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["valid"] is True and d["value"] > 0 and d["scaling"] == "strong"
    assert d["histogram"].startswith("counted 1/2 per rank")          # (forced at this size: every build started with dist.plan_sharded)
    # (two ranks over gloo are not an RCCL communicator, and the line says so)
    assert d["config"]["ranks"] == 2 and d["config"]["rccl_ranks"] is None and d["config"]["backend"] == "gloo" and d["latency_ms"] > 0 and d["exchange_ms"] >= 0
    assert "1 GiB" not in d["metric"] and d["config"]["index"] == "int32"
    assert len(d["per_rank"]["sort_ms"]) == 2 and sum(d["per_rank"]["rows"]) == (1 << 24) + 1
    # `value` is the latency of complete builds (sort + exchange, nothing overlapped); the pipelined rate is a secondary field
    assert d["config"]["pipelined"] is False and d["pipelined"]["MBps"] > 0 and d["ms_per_step"] >= 0.5 * d["latency_ms"]


@pytest.mark.parametrize("world,workload", [(4, "random"), (4, "dna"), (8, "dna_tandem")])      # (4-rank tandem DNA: int64 rows and the sharded forward BWT below)
def test_bench_four_and_eight_ranks_one_gpu(world, workload):
    """First-contact hardening of the 4- and 8-rank flows (the driver's 8-GPU node is the first place they meet RCCL): every rank
    of `python bench.py --gpus N` shares cuda:0 over gloo.  random: eight even key ranges; dna: 16 two-byte keys in all, so the
    cuts fall INSIDE heavy keys (deeper histogram, 4-byte-prefix ranges); dna_tandem: the shards stop unresolved and finish with
    the distributed prefix doubling.  The assembled array is checked on the device by rank 0 (`valid`)."""
    env = dict(os.environ, MSUFSORT_BENCH_BACKEND="gloo", MSUFSORT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", MSUFSORT_DIST_SHARDED_HIST="1")
    n = 1 << 21 if workload == "dna_tandem" else 1 << 23          # (a dozen doubling steps through 8 python ranks: 80 s at 16 MiB)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0",
           "--size", str(n), "--workload", workload, "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == world and d["config"]["ranks"] == world and d["valid"] is True
    rows = d["per_rank"]["rows"]
    assert len(rows) == world and sum(rows) == n + 1
    if workload == "random":
        assert d["histogram"].startswith(f"counted 1/{world} per rank")      # the histogram all-reduced, the stripe sums all-gathered
    if workload in ("random", "dna"):
        assert max(rows) <= 1.25 * (n / world) + 2, rows          # balanced, also where two-byte keys are heavier than a shard
    if workload == "dna_tandem":
        assert d["doubling"]["doubling_steps"] >= 1 and d["doubling"]["updates"] > 0


def test_bench_refuses_more_ranks_than_gpus():
    """Without the one-device test hook `--gpus N` on a box with fewer GPUs must fail loudly, not measure one GPU."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("MSUFSORT_BENCH_ONE_DEVICE", "WORLD_SIZE", "RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                        "--size", str(1 << 20), "--no-cpu"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_under_torchrun():
    """The other launch form (the ranks started by torch.distributed.run around bench.py) still works."""
    env = dict(os.environ, MSUFSORT_BENCH_BACKEND="gloo", MSUFSORT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--size", str(1 << 22), "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(lines[-1])["n_gpus"] == 2


def test_bench_two_ranks_deep_ties():
    """Same flow on DNA with planted tandem repeats: the shards stop unresolved and finish with the distributed prefix
    doubling (each rank sorts its own groups, rank updates are all-gathered once per step)."""
    env = dict(os.environ, MSUFSORT_BENCH_BACKEND="gloo", MSUFSORT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--size", str(3 << 20), "--workload", "dna_tandem", "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["valid"] is True and d["doubling"]["doubling_steps"] >= 1 and d["doubling"]["updates"] > 0


def test_cpp_dropin_header_builds_and_runs(tmp_path):
    exe = str(tmp_path / "demo")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "demo.cpp"),
                    "-L" + os.path.join(ROOT, "msufsort_amd", "lib"), "-lmsufsort_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "msufsort_amd", "lib"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SA[0] = 11, SA[1] = 10" in r.stdout and "sentinel row = 5, round trip ok" in r.stdout


def test_demo_cli_modes(tmp_path):
    """CLI parity with the reference demo (reference main.cpp:290-502): modes s, l, b on a file, t self test."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from msufsort_amd import gen
    exe = str(tmp_path / "msufsort_demo")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "msufsort_demo.cpp"),
                    "-L" + os.path.join(ROOT, "msufsort_amd", "lib"), "-lmsufsort_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "msufsort_amd", "lib"), "-o", exe], check=True)
    f = tmp_path / "in.bin"
    gen.text_bytes(200000, 17).tofile(f)
    for mode, must in (("s", "suffix array validated"), ("l", "lcp array validated"), ("b", "BWT validated")):
        r = subprocess.run([exe, mode, str(f), "4"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and must in r.stdout, r.stdout + r.stderr
    # 40 MiB: the drop-in header takes msufsort_hip_make_sa_multi (all visible GPUs, slices streamed to the host) from 32 MiB on
    big = tmp_path / "big.bin"
    gen.random_bytes(40 << 20, 5).tofile(big)
    r = subprocess.run([exe, "s", str(big), "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "suffix array validated" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "t"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 errors" in r.stdout, r.stdout + r.stderr


def test_reference_demo_runs_on_the_dropin(tmp_path):
    """The reference's own demo executable - main.cpp compiled UNCHANGED against include/ and linked to libmsufsort_hip.so by
    oracle/Makefile in the build container (oracle/_ref/msufsort_demo_dropin) - runs its s / b / l modes on the GPU engine and
    its OWN validators (main.cpp:236-270 adjacent-pair check, the BWT round trip main.cpp:470-487) accept the results."""
    sys.path.insert(0, ROOT)
    from msufsort_amd import gen
    exe = os.path.join(ROOT, "oracle", "_ref", "msufsort_demo_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/msufsort_demo_dropin was not built (no reference tree when the snapshot was made)")
    for name, data in (("text", gen.text_bytes(3 << 20, 23)), ("random", gen.random_bytes(40 << 20, 9))):          # 40 MiB: the streaming entry point
        f = tmp_path / (name + ".bin")
        data.tofile(f)
        for mode in ("s", "b") if name == "random" else ("s", "b", "l"):
            r = subprocess.run([exe, mode, str(f), "8"], capture_output=True, text=True, timeout=900)
            assert r.returncode == 0 and "caught exception" not in r.stdout and "ERROR" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
            if mode != "l":
                assert "test completed and results validated successfully" in r.stdout, r.stdout[-2000:]
            else:
                assert "suffix array completed" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("world", [2, 4])
def test_bench_two_stage_sharded_text(world):
    """Text over several ranks the reference's way: B* suffixes sorted by key-range shards, the sorted-B* slices all-gathered (a
    third of the suffix array), every rank induces the rest - the whole array is on every rank, checked on the device by rank 0."""
    env = dict(os.environ, MSUFSORT_BENCH_BACKEND="gloo", MSUFSORT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    n = 1 << 23
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0",
           "--size", str(n), "--workload", "text", "--two-stage", "1", "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == world and d["valid"] is True
    assert d["two_stage_sharded"]["two_stage_status"] == 0 and d["allgatherv_bytes_per_rank"] < 4 * n * 0.5


def _bench(args, timeout=900, env_extra=None):
    env = dict(os.environ, MSUFSORT_BENCH_BACKEND="gloo", MSUFSORT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *[str(a) for a in args]], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("world,workload,n", [(4, "dna_tandem", 1 << 21), (2, "random", 1 << 23), (2, "text", 1 << 21), (8, "dna", 1 << 22)])
def test_bench_int64_rows_multi_process(world, workload, n):
    """BASELINE config 5 as it is written - int64 rows, one process per GPU, the wide engine's shards, the 16-byte-update
    distributed doubling, the all-gatherv of 8-byte rows - at sizes the CPU checker finishes (`--index int64` forces what
    n > 2^31 - 2 selects by itself): every row equal to the unmodified reference's (--check-reference) and accepted by the 64-bit
    on-device checker.  dna_tandem: the shards stop unresolved, the doubling runs (small exchange windows: several per step)."""
    d = _bench(["--gpus", world, "--steps", 1, "--warmup", 0, "--size", n, "--workload", workload, "--index", "int64", "--no-cpu", "--check-reference"],
               env_extra={"MSUFSORT_DIST_WINDOW": "20000"} if workload == "dna_tandem" else None)
    assert d["n_gpus"] == world and d["valid"] is True and d["config"]["index"] == "int64" and "reference" in d["valid_against"]
    assert sum(d["per_rank"]["rows"]) == n + 1 and d["allgatherv_bytes_per_rank"] == int(8 * (n + 1) * (world - 1) / world)
    if workload == "dna_tandem":
        db = d["doubling"]
        assert db["index_bytes"] == 8 and db["doubling_steps"] >= 2 and db["updates"] > 0 and db["windows"] > db["doubling_steps"]


@pytest.mark.parametrize("world,workload,index", [(2, "random", "int32"), (4, "random", "int64"), (4, "dna_tandem", "int32"), (2, "dna_tandem", "int64"), (2, "text", "int32")])
def test_bench_sharded_forward_bwt(world, workload, index):
    """The forward transform over several ranks with the BYTES exchanged instead of the rows (reference semantics
    msufsort.cpp:1771-1817): bytes + sentinel row equal to the reference's, for sort-all shards (random), shards that need the
    distributed doubling (dna_tandem) and the two-stage sharded text build (every rank already holds all rows: no exchange)."""
    n = 1 << 21 if workload == "dna_tandem" else 1 << 22
    extra = ["--two-stage", 1] if workload == "text" else []
    d = _bench(["--gpus", world, "--steps", 1, "--warmup", 0, "--size", n, "--workload", workload, "--index", index, "--op", "sa,fbwt", "--no-cpu", "--check-reference", *extra])
    assert d["valid"] is True and "BWT" in d["valid_against"] and d["ops_ms"]["fbwt"] > 0
    fb = d["forward_bwt"]
    assert 1 <= fb["sentinel_row"] <= n
    if workload == "text":
        assert d["two_stage_sharded"]["two_stage_status"] == 0 and "nothing beyond" in fb["exchanged"]
    else:
        assert fb["bwt_bytes_received"] <= n * (world - 1) / world * 1.3 + 2


def test_bench_refuses_when_hbm_is_short():
    """n = 2^33 with int64 rows needs ~220 GiB per rank at 8 ranks - and far more at 2: bench.py says what it needs and stops
    before allocating (here: 2 ranks sharing the one GPU)."""
    env = dict(os.environ, MSUFSORT_BENCH_BACKEND="gloo", MSUFSORT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", str(1 << 33), "--workload", "dna", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode != 0 and "GiB of HBM per GPU" in r.stderr and "rank replica" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("args", [["--workload", "random", "--size", 1 << 22], ["--workload", "dna_tandem", "--size", 1 << 21, "--index", "int64", "--op", "sa,fbwt"],
                                  ["--workload", "text", "--size", 1 << 22, "--two-stage", 1]])
def test_bench_dist_path_on_rccl_with_one_rank(args):
    """RCCL refuses two ranks on one device, so no box this repository has met could run the N > 1 flow on the backend it is written
    for.  What a one-GPU box CAN do: one rank, backend nccl (= RCCL), forced through the N > 1 code path - process group with
    device_id, broadcast of the text, the all-reduces (int32 / int64 / float64, MAX and SUM), barrier, the two-stage exchange
    callback, the distributed doubling's window loop with its collectives, the byte-slice exchange of the forward BWT - everything
    but the point-to-point sends (a one-rank all-gatherv posts none).  API misuse against RCCL shows here, not on the 8-GPU node."""
    env = {k: v for k, v in os.environ.items() if k not in ("MSUFSORT_BENCH_BACKEND", "MSUFSORT_BENCH_ONE_DEVICE")}
    env.update(MSUFSORT_BENCH_FORCE_DIST="1", MSUFSORT_DIST_ALWAYS_COLLECTIVE="1", MSUFSORT_DIST_WINDOW="50000", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MSUFSORT_DIST_SHARDED_HIST="1")          # (the histogram's all-reduce and all-gather on a communicator of their own, through RCCL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29633",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu", "--check-reference", *[str(a) for a in args]]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[-1])
    assert d["valid"] is True and d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1 and d["n_gpus"] == 1
    if args[1] == "random":
        assert d["histogram"].startswith("counted 1/1 per rank")      # hist_part -> all-reduce -> hist_plan -> all-gather -> hist_install, on RCCL


@pytest.mark.parametrize("window", [0, 1 << 23])
def test_bench_int64_rows_at_a_size_that_needs_them_on_rccl(window):
    """BASELINE config 5's own code path (`bench.py --gpus N --index auto` -> dist.build_sa_sharded(index_bytes=8), ShardState,
    _replicate_ranks, _distributed_doubling, forward_bwt_sharded) at a size that NEEDS int64 rows: n = 2^31 + 12,345 bytes of
    tandem-repeat DNA, ONE rank on RCCL forced through the N > 1 flow with every collective issued - nothing travels between ranks, but
    every offset, window count, update and bounds[] entry of the Python flow is beyond 32 bits for real (the reference's index is
    int32 with two flag bits: msufsort.h:47, 84-93).  Checked on the device: the 64-bit checker, every row equal to the single-process
    build's (msufsort_hip_make_sa_i64_dev, the path test_gpu_big.py / test_gpu_parity.py pin), BWT bytes + sentinel row likewise.
    window = 0: the default exchange windows (2^25 updates, 2^26 group heads: dozens per step at this size); 2^23: four times as many."""
    import gc

    import torch
    sys.path.insert(0, ROOT)
    from msufsort_amd import _lib
    gc.collect()
    torch.cuda.empty_cache()
    _lib.lib().msufsort_hip_release_cached()          # (what earlier tests of this process left in the one-shot entry points' contexts)
    free, _ = torch.cuda.mem_get_info()
    if free < 235 << 30:
        pytest.skip(f"needs ~225 GiB of free HBM ({free >> 30} GiB free)")
    n = (1 << 31) + 12345
    env = {k: v for k, v in os.environ.items() if k not in ("MSUFSORT_BENCH_BACKEND", "MSUFSORT_BENCH_ONE_DEVICE", "MSUFSORT_DIST_WINDOW")}
    env.update(MSUFSORT_BENCH_FORCE_DIST="1", MSUFSORT_DIST_ALWAYS_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if window:
        env["MSUFSORT_DIST_WINDOW"] = str(window)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu", "--check-single", "--workload", "dna_tandem", "--size", str(n),
           "--index", "auto", "--op", "sa,fbwt"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[-1])
    assert d["valid"] is True and "single-process" in d["valid_against"] and "BWT" in d["valid_against"]
    assert d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1 and d["config"]["index"] == "int64" and d["config"]["n"] == n
    db = d["doubling"]
    assert db["index_bytes"] == 8 and db["doubling_steps"] >= 2 and db["updates"] > (1 << 28)
    assert db["windows"] >= db["doubling_steps"] + (32 if window else 8)            # several exchange windows in the steps that have work
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"r06_one_rank_rccl_int64_2pow31_window{window}.json"), "w") as f:
            f.write(lines[-1] + "\n")


@pytest.mark.parametrize("world,k,args", [(2, 3, ["--workload", "random", "--size", 1 << 24]), (2, 2, ["--workload", "dna_tandem", "--size", 1 << 22, "--index", "int64", "--op", "sa,fbwt"]),
                                          (4, 2, ["--workload", "dna", "--size", 1 << 23])])
def test_bench_sub_shards_overlap_the_exchange(world, k, args):
    """Round 6 (reference: threads leave their partitions while others still sort, msufsort.cpp:1652-1683): every rank sorts its key
    range as k sub-shards (one plan: msufsort_hip_opts.reuse_plan, or the sharded histogram's stripe sums installed per sub-shard) and
    posts sub-slice j as soon as it is sorted - grouped sends / receives that travel while sub-shard j + 1 is sorted.  Rows (and
    the BWT) equal to the unmodified reference's; cuts inside heavy keys (DNA); shards that stop unresolved (tandem DNA, int64
    rows): group heads moved to the rank's slice, distributed doubling."""
    d = _bench(["--gpus", world, "--steps", 1, "--warmup", 0, "--no-cpu", "--check-reference", *args],
               env_extra={"MSUFSORT_DIST_SUBSHARDS": str(k), "MSUFSORT_DIST_SHARDED_HIST": "1" if args[1] == "random" else "0"})
    assert d["valid"] is True and "reference" in d["valid_against"] and d["config"]["overlap"]["sub_shards_per_rank"] == k
    assert len(d["per_rank"]["rows"]) == world
    if args[1] == "random":
        assert d["histogram"].startswith(f"counted 1/{world} per rank")
    if args[1] == "dna_tandem":
        assert d["doubling"]["doubling_steps"] >= 1
