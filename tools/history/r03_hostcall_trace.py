"""Five movi_pml_host calls at the command's chunk size with touched pageable buffers (walk only: NULL vector), to be run under
`rocprofv3 --hip-trace --stats`: which HIP calls make up the ~6 ms per call?"""
import sys, os, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import movi_amd
from movi_amd._lib import lib, check, QueryStatsC
d = "/tmp/movi_bench_cache/pg_5000000_64_0.001_11_m6"
rf = sorted(glob.glob(d + "/reads_*x150_*.bin"))[0]
n = 223696
bases = np.fromfile(rf, np.uint8, count=n * 150)
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
err = np.zeros(n, np.uint8); err[:] = 0
out = np.zeros(n * 150, np.uint16); out[:] = 1
ix = movi_amd.MoveIndex.load(d)
st = QueryStatsC()
for mode in ("walk only", "vector down"):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        check(lib().movi_pml_host(ix._h, bases.ctypes.data, offs.ctypes.data, n, out.ctypes.data if mode == "vector down" else None, err.ctypes.data, C.byref(st)))
        ts.append(time.perf_counter() - t0)
    print("%-12s per call: %s ms" % (mode, " ".join("%.2f" % (t * 1e3) for t in ts)))
