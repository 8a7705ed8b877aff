# round 6: movi_pml_host (masks down + host expansion), worker threads of the pool -- one process per setting (the pool only grows)
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import movi_amd
from movi_amd._lib import QueryStatsC, check, lib
thr = int(sys.argv[1])
D = ".bench_cache/pg_5000000_64_0.001_11_m6"
idx = movi_amd.MoveIndex.load(D)
bases = np.fromfile(D + "/reads_1000000x150_0.01.bin", np.uint8)
n = bases.size // 150
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
st = QueryStatsC()
hb = movi_amd.pinned_empty(bases.size, np.uint8); hb[:] = bases
ho = movi_amd.pinned_empty(bases.size, np.uint16)
idx.set_option("host_threads", thr)
ts = []
for _ in range(12):
    t0 = time.perf_counter()
    check(lib().movi_pml_host(idx._h, hb.ctypes.data, offs.ctypes.data, n, ho.ctypes.data, None, C.byref(st)))
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print("host_threads %2d: best %.2f median %.2f worst %.2f Gbases/s" % (thr, bases.size / ts[0] / 1e9, bases.size / ts[len(ts) // 2] / 1e9, bases.size / ts[-1] / 1e9), flush=True)
