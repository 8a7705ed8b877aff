"""Steady-state rate of the C-ABI host entry points (host buffers in, host buffers out; result buffer allocated and
touched once, as a long-running caller would): pageable buffers (synchronous path) against page-locked ones (overlapped
path: chunks in flight on three streams), next to the raw copy rates of the same volumes and the cost of page-locking."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import movi_amd
from movi_amd._lib import lib, QueryStatsC, check
from tools import synth
six = synth.synth_index(10_000_000, mode=6, seed=1)
ix = movi_amd.MoveIndex.from_image(six.image())


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts[0], min(ts[1:])


for n, L in ((1_000_000, 150), (100_000, 10_000)):
    bases, offs = synth.synth_reads(six, n, L, seed=2, sub_rate=0.01 if L < 1000 else 0.08, n_rate=0.001)   # BASELINE configs 2 / 3
    out = np.ones(bases.size, np.uint16)                     # touched
    err = np.zeros(n, np.uint8)
    m, c = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    a, b, s = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint64)
    st = QueryStatsC()
    t0 = time.perf_counter()
    pb = movi_amd.pinned_empty(bases.size, np.uint8)
    pout = movi_amd.pinned_empty(bases.size, np.uint16)
    t_alloc = time.perf_counter() - t0
    pb[:] = bases
    pout[:] = 1
    print("== %d x %d bp: movi_host_alloc of %.2f GB: %.3f s = %.1f GB/s" % (n, L, 3 * bases.size / 1e9, t_alloc, 3 * bases.size / t_alloc / 1e9))
    t0 = time.perf_counter()
    check(lib().movi_host_register(out.ctypes.data, out.nbytes))
    t_reg = time.perf_counter() - t0
    check(lib().movi_host_unregister(out.ctypes.data))
    print("   movi_host_register of a touched %.2f GB buffer: %.3f s = %.1f GB/s" % (out.nbytes / 1e9, t_reg, out.nbytes / t_reg / 1e9))
    for name, hb, ho in (("pageable", bases, out), ("page-locked", pb, pout)):
        f, r = best(lambda: check(lib().movi_pml_host(ix._h, hb.ctypes.data, offs.ctypes.data, n, ho.ctypes.data, err.ctypes.data, C.byref(st))))
        print("   movi_pml_host, %s: first call %.3f s, best of the next four %.4f s = %.2f Gbases/s" % (name, f, r, bases.size / r / 1e9))
        f, r = best(lambda: check(lib().movi_zml_host(ix._h, hb.ctypes.data, offs.ctypes.data, n, ho.ctypes.data, err.ctypes.data, C.byref(st))), 3)
        print("   movi_zml_host, %s: best %.4f s = %.2f Gbases/s" % (name, r, bases.size / r / 1e9))
        f, r = best(lambda: check(lib().movi_count_host(ix._h, hb.ctypes.data, offs.ctypes.data, n, m.ctypes.data, c.ctypes.data, err.ctypes.data, C.byref(st))))
        print("   movi_count_host, %s: best %.4f s = %.2f Gbases/s" % (name, r, bases.size / r / 1e9))
        f, r = best(lambda: check(lib().movi_pml_classify_host(ix._h, hb.ctypes.data, offs.ctypes.data, n, 150, 8, a.ctypes.data, b.ctypes.data, s.ctypes.data, err.ctypes.data, C.byref(st))))
        print("   movi_pml_classify_host, %s: best %.4f s = %.2f Gbases/s" % (name, r, bases.size / r / 1e9))
    assert (pout == out).all()
    d_in = torch.empty(bases.size, dtype=torch.uint8, device="cuda"); d_out = torch.empty(bases.size, dtype=torch.int16, device="cuda")
    for nm, hi, ho in (("pageable", torch.from_numpy(bases), torch.from_numpy(out.view(np.int16))),
                       ("page-locked", torch.from_numpy(pb), torch.from_numpy(pout.view(np.int16)))):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); d_in.copy_(hi); torch.cuda.synchronize(); t1 = time.perf_counter(); ho.copy_(d_out); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("   %s copies of the same buffers: H2D %.1f GB/s (%.1f ms), D2H %.1f GB/s (%.1f ms)" % (nm, bases.size / (t1 - t0) / 1e9, (t1 - t0) * 1e3, 2 * bases.size / (t2 - t1) / 1e9, (t2 - t1) * 1e3))
    del pb, pout
