"""Steady-state rate of the C-ABI host entry point movi_pml_host (host buffers in, host buffers out; result buffer
allocated and touched once, as a long-running caller would) next to the raw copy rates of the same volumes."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import movi_amd
from movi_amd._lib import lib, QueryStatsC
from tools import synth
six = synth.synth_index(10_000_000, mode=6, seed=1)
ix = movi_amd.MoveIndex.from_image(six.image())
for n, L in ((1_000_000, 150), (100_000, 10_000)):
    bases, offs = synth.synth_reads(six, n, L, seed=2, sub_rate=0.01, n_rate=0.001)
    out = np.ones(bases.size, np.uint16)                     # touched
    err = np.zeros(n, np.uint8)
    st = QueryStatsC()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        rc = lib().movi_pml_host(ix._h, bases.ctypes.data, offs.ctypes.data, n, out.ctypes.data, err.ctypes.data, C.byref(st))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    print("movi_pml_host %d x %d: first call %.3f s, best of the next four %.3f s = %.2f Gbases/s" % (n, L, ts[0], min(ts[1:]), bases.size / min(ts[1:]) / 1e9))
    d_in = torch.empty(bases.size, dtype=torch.uint8, device="cuda"); d_out = torch.empty(bases.size, dtype=torch.int16, device="cuda")
    h_in = torch.from_numpy(bases); h_out = torch.from_numpy(out.view(np.int16))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d_in.copy_(h_in); torch.cuda.synchronize(); t1 = time.perf_counter(); h_out.copy_(d_out); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("  pageable copies of the same buffers: H2D %.1f GB/s (%.1f ms), D2H %.1f GB/s (%.1f ms)" % (bases.size / (t1 - t0) / 1e9, (t1 - t0) * 1e3, 2 * bases.size / (t2 - t1) / 1e9, (t2 - t1) * 1e3))
