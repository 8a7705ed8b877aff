import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import numpy as np
import movi_amd
from oracle.oracle import Oracle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
img = open(os.path.join(ROOT, "tests/golden/index_regular-thresholds/index.movi"), "rb").read()
gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
lines = open(os.path.join(ROOT, "tests/golden/sample.fastq"), "rb").read().split(b"\n")
reads = [lines[i + 1].strip() for i in range(0, len(lines) - 3, 4)][:4] + [b"A", b"ACGT", b"TTTTTTTT"]
lens = [len(r) for r in reads]
offs = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
bases = np.frombuffer(b"".join(reads), np.uint8)
out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
print("rc", rc, st, err)
for i, r in enumerate(reads):
    e = cpu.pml(r)
    g = out[int(offs[i]):int(offs[i+1])]
    d = np.flatnonzero(e != g)
    print(i, len(r), "first diff", d[:1], "gpu", g[:12], "cpu", e[:12])
