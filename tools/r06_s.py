# round 6: movi_pml_host, both ways down -- chunk size sweep ("pipe_chunk_bases"); 1 M x 150 bp on the c2 index, page-locked buffers
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import movi_amd
from movi_amd._lib import QueryStatsC, check, lib
D = ".bench_cache/pg_5000000_64_0.001_11_m6"
idx = movi_amd.MoveIndex.load(D)
bases = np.fromfile(D + "/reads_1000000x150_0.01.bin", np.uint8)
n = bases.size // 150
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
st = QueryStatsC()
pin = movi_amd.pinned_empty
hb = pin(bases.size, np.uint8); hb[:] = bases
ho = pin(bases.size, np.uint16)
def run(hm, share, chunk):
    idx.set_option("host_masks", hm); idx.set_option("host_mask_share", share); idx.set_option("pipe_chunk_bases", chunk)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        check(lib().movi_pml_host(idx._h, hb.ctypes.data, offs.ctypes.data, n, ho.ctypes.data, None, C.byref(st)))
        ts.append(time.perf_counter() - t0)
    return bases.size / min(ts[1:]) / 1e9, bases.size / sorted(ts[1:])[3] / 1e9
from movi_amd.engine import mask_words
hw = np.zeros(mask_words(n, bases.size), np.uint32)
def masks_only():
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        check(lib().movi_pml_mask_host(idx._h, hb.ctypes.data, offs.ctypes.data, n, hw.ctypes.data, None, C.byref(st)))
        ts.append(time.perf_counter() - t0)
    return bases.size / min(ts[1:]) / 1e9
for rep in range(3):
    print("masks only %.2f | vector through masks %.2f (median %.2f) | vector by DMA %.2f" % ((masks_only(),) + run(-1, 70, 0) + (run(0, 70, 0)[0],)), flush=True)
