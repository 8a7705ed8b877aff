import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import movi_amd
from movi_amd._lib import lib, QueryStatsC, check
from tools import synth
six = synth.synth_index(10_000_000, mode=6, seed=1)
ix = movi_amd.MoveIndex.from_image(six.image())
n, L = 1_000_000, 150
bases, offs = synth.synth_reads(six, n, L, seed=2, sub_rate=0.01, n_rate=0.001)
pb = movi_amd.pinned_empty(bases.size, np.uint8); pb[:] = bases
pout = movi_amd.pinned_empty(bases.size, np.uint16); pout[:] = 1
err = np.zeros(n, np.uint8); st = QueryStatsC()
for rep in range(3):
    t0 = time.perf_counter()
    check(lib().movi_pml_host(ix._h, pb.ctypes.data, offs.ctypes.data, n, pout.ctypes.data, err.ctypes.data, C.byref(st)))
    print("call %d: %.2f ms" % (rep, (time.perf_counter() - t0) * 1e3))
