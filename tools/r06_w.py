# round 6: do the walks of several small batches run side by side when they are enqueued on different streams?  (62.5 k x 150 bp each, c2's index)
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import movi_amd
D = ".bench_cache/pg_5000000_64_0.001_11_m6"
idx = movi_amd.MoveIndex.load(D)
dev = torch.device("cuda", 0)
bases = np.fromfile(D + "/reads_1000000x150_0.01.bin", np.uint8)
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500
K = 8
streams = [torch.cuda.Stream() for _ in range(K)]
bufs = []
for k in range(K):
    b = torch.from_numpy(bases[k * nr * 150:(k + 1) * nr * 150].copy()).to(dev)
    o = torch.from_numpy((np.arange(nr + 1, dtype=np.int64) * 150)).to(dev)
    out = torch.empty(nr * 150, dtype=torch.int16, device=dev)
    err = torch.zeros(nr, dtype=torch.uint8, device=dev)
    bufs.append((b, o, out, err))
def call(k, s):
    b, o, out, err = bufs[k]
    idx.pml_device(b.data_ptr(), o.data_ptr(), nr, nr * 150, out.data_ptr(), err.data_ptr(), s.cuda_stream, 0)
for k in range(K): call(k, streams[k])
torch.cuda.synchronize()
for mode in ("one stream", "own streams"):
    for n in (1, 2, 3, 4, 8):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n): call(k, streams[0] if mode == "one stream" else streams[k])
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("%-11s %d batches: %.3f ms (%.1f Gbases/s)" % (mode, n, min(ts) * 1e3, n * nr * 150 / min(ts) / 1e9), flush=True)
