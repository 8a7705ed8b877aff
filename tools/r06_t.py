# round 6: one movi_pml_mask_host / movi_pml_host call under rocprofv3 --memory-copy-trace --kernel-trace: where a call's 4 ms go
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import movi_amd
from movi_amd._lib import QueryStatsC, check, lib
from movi_amd.engine import mask_words
D = ".bench_cache/pg_5000000_64_0.001_11_m6"
idx = movi_amd.MoveIndex.load(D)
bases = np.fromfile(D + "/reads_1000000x150_0.01.bin", np.uint8)
n = bases.size // 150
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
st = QueryStatsC()
hb = movi_amd.pinned_empty(bases.size, np.uint8); hb[:] = bases
hw = np.zeros(mask_words(n, bases.size), np.uint32)
ho = movi_amd.pinned_empty(bases.size, np.uint16)
for which in ("mask", "vector_via_masks"):
    for rep in range(4):
        t0 = time.perf_counter()
        if which == "mask":
            check(lib().movi_pml_mask_host(idx._h, hb.ctypes.data, offs.ctypes.data, n, hw.ctypes.data, None, C.byref(st)))
        else:
            check(lib().movi_pml_host(idx._h, hb.ctypes.data, offs.ctypes.data, n, ho.ctypes.data, None, C.byref(st)))
        print(which, rep, "%.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    time.sleep(0.05)
