import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import movi_amd
from tools import synth
six = synth.synth_index(10_000_000, mode=6, seed=1)
ix = movi_amd.MoveIndex.from_image(six.image())
for n, L in ((1_000_000, 150), (33_000, 10_000)):
    bases, offs = synth.synth_reads(six, n, L, seed=2, sub_rate=0.01, n_rate=0.001)
    for rep in range(3):
        t0 = time.perf_counter(); out, st = ix.query_pml_packed(bases, offs); dt = time.perf_counter() - t0
    print("movi_pml_host %d x %d: %.3f s  %.2f Gbases/s" % (n, L, dt, bases.size / dt / 1e9))
    # raw copy rates for the same volumes
    d_in = torch.empty(bases.size, dtype=torch.uint8, device="cuda"); d_out = torch.empty(bases.size, dtype=torch.int16, device="cuda")
    h_in = torch.from_numpy(bases); h_out = torch.empty(bases.size, dtype=torch.int16)
    for name, hi, ho in (("pageable", h_in, h_out), ("pinned", h_in.pin_memory(), h_out.pin_memory())):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d_in.copy_(hi); torch.cuda.synchronize(); t1 = time.perf_counter(); ho.copy_(d_out); torch.cuda.synchronize(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t0 = time.perf_counter(); d_in.copy_(hi); torch.cuda.synchronize(); t1 = time.perf_counter(); ho.copy_(d_out); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("  %s: H2D %.1f GB/s, D2H %.1f GB/s" % (name, bases.size / (t1 - t0) / 1e9, 2 * bases.size / (t2 - t1) / 1e9))
    t0 = time.perf_counter(); x = h_out.numpy().copy(); print("  host memcpy 1 thread: %.1f GB/s" % (x.nbytes / (time.perf_counter() - t0) / 1e9))
