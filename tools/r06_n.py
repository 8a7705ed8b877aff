# round 6: movi_pml_host with both ways down side by side (mixed) -- share sweep; 1 M x 150 bp on the c2 index
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import movi_amd
from movi_amd._lib import QueryStatsC, check, lib
D = ".bench_cache/pg_5000000_64_0.001_11_m6"
idx = movi_amd.MoveIndex.load(D) if hasattr(movi_amd.MoveIndex, "load") else movi_amd.MoveIndex.from_image(open(D + "/index.movi", "rb").read())
bases = np.fromfile(D + "/reads_1000000x150_0.01.bin", np.uint8)
n = bases.size // 150
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
st = QueryStatsC()
def run(mk, hm, share, threads=0):
    idx.set_option("host_masks", hm); idx.set_option("host_mask_share", share); idx.set_option("host_threads", threads)
    hb = mk(bases.size, np.uint8); hb[:] = bases
    ho = mk(bases.size, np.uint16); ho[:] = 0xFFFF
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        check(lib().movi_pml_host(idx._h, hb.ctypes.data, offs.ctypes.data, n, ho.ctypes.data, None, C.byref(st)))
        ts.append(time.perf_counter() - t0)
    return bases.size / min(ts[1:]) / 1e9, bases.size / sorted(ts[1:])[2] / 1e9, ho
pin = movi_amd.pinned_empty
page = lambda k, dt: np.empty(k, dt)
_, _, ref = run(pin, 0, 40)
for rep in range(2):
    for taper in (0, 1):
        idx.set_option("pipe_taper", taper)
        print("taper %d: page-locked vector %.2f | masks %.2f" % (taper, run(pin, 0, 40)[0], run(pin, 1, 40)[0]), end="")
        b, m, ho = run(pin, -1, 40)
        print(" | both ways, chosen chunk by chunk %.2f (median %.2f) ok %s" % (b, m, bool((ho == ref).all())), end="")
        idx.set_option("pml_via_mask", 0)
        print(" | dealt 40 %%: %.2f  50 %%: %.2f" % (run(pin, 2, 40)[0], run(pin, 2, 50)[0]), end="")
        idx.set_option("pml_via_mask", -1)
        b, m, ho = run(page, -1, 40)
        print(" | pageable default %.2f (median %.2f) ok %s" % (b, m, bool((ho == ref).all())))
