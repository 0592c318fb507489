"""the unmodified reference (oracle/_ref) on the host cores of the GPU box: SA of 256 MiB random bytes with 16 .. all hardware threads
(its workers spin-wait: more threads than free cores can be slower than fewer); every run in a child with a watchdog"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = 1 << 28
for th in (16, 32, 64, 96, 128, 192, os.cpu_count()):
    r = bench.cpu_reference_runs("random", 12345, n, [(th, "sa")], 90)
    print(th, r[0] if r else "did not finish within 90 s", flush=True)
