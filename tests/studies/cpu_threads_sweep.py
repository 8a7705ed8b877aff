"""How the CPU port (oracle/movi_oracle.c, the cpu_baseline of bench.py) scales with threads on this host:
nproc, the cgroup CPU quota and Gbases/s at 1 ... os.cpu_count() threads on the random 10 M-row table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tools import synth
from oracle.oracle import Oracle
print("os.cpu_count() =", os.cpu_count(), " sched_getaffinity =", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, "=", open(f).read().strip())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|L3' ")
six = synth.synth_index(10_000_000, mode=6, seed=20260529)
cpu = Oracle(six.image())
bases, offs = synth.synth_reads(six, 400_000, 150, seed=3, sub_rate=0.01, n_rate=0.001)
T = os.cpu_count() or 1
ts = [t for t in (1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256, 384, 512) if t <= 2 * T]
for t in ts:
    n = min(400_000, 20_000 * t)
    sb, so = bases[: int(offs[n])], offs[: n + 1]
    cpu.pml_batch(sb, so, threads=t, strands=16)
    t0 = time.perf_counter()
    cpu.pml_batch(sb, so, threads=t, strands=16)
    dt = time.perf_counter() - t0
    print("threads %4d: %8.4f Gbases/s (%.1f Mbases/s per thread)" % (t, sb.size / dt / 1e9, sb.size / dt / 1e6 / t), flush=True)
