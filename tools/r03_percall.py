"""Per-call cost of the *_host entry points at the command's chunk size (2^25 bases of 150 bp reads, pageable buffers):
where do the ~6 ms per GPU call of `movi query --no-output` go?  First call (staging allocations, top-of-walk table) vs steady state."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import movi_amd
d = "/tmp/movi_bench_cache/pg_5000000_64_0.001_11_m6"
import glob
rf = sorted(glob.glob(d + "/reads_*x150_*.bin"))[0]
n = 223696
bases = np.fromfile(rf, np.uint8, count=n * 150)
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
t0 = time.perf_counter(); ix = movi_amd.MoveIndex.load(d); print("index load %.4f s" % (time.perf_counter() - t0))
for name, fn in (("classify (no vector)", lambda: ix.classify_packed(bases, offs, 150, 1)), ("pml (vector down)", lambda: ix.query_pml_packed(bases, offs)), ("count", lambda: ix.query_count_packed(bases, offs))):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print("%-22s per call: %s ms" % (name, " ".join("%.2f" % (t * 1e3) for t in ts)))
