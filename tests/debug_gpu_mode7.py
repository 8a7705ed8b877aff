"""Debug helper (GPU box): mode 7 count / ZML mismatches against the oracle, printed read by read."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import movi_amd
from oracle import build_index as B
from oracle.oracle import Oracle

ref = B.read_fasta(os.path.join(ROOT, "tests", "golden", "ref.fasta"))[0][1]
img = B.build_index_from_seqs([ref], 7)
gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
rng = np.random.default_rng(1)
reads = []
for _ in range(300):
    L = int(rng.integers(1, 200))
    s = int(rng.integers(0, len(ref) - L))
    r = bytearray(ref[s:s + L])
    for k in range(L):
        if rng.random() < 0.02:
            r[k] = b"ACGTN"[rng.integers(0, 5)]
    reads.append(bytes(r))
reads += [b"A", b"C", b"G", b"T", b"AC", b"TTTT"]
lens = [len(r) for r in reads]
offs = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
bases = np.frombuffer(b"".join(reads), np.uint8)
m, c, st = gpu.query_count_packed(bases, offs)
em, ec = cpu.count_batch(bases, offs, threads=4)
bad = np.flatnonzero((m != em) | (c != ec))
print("count mismatches:", bad.size, "of", len(reads), "stats", st)
for i in bad[:12]:
    print(i, len(reads[i]), "gpu", int(m[i]), int(c[i]), "cpu", int(em[i]), int(ec[i]), reads[i][-12:])
z, _ = gpu.query_zml_packed(bases, offs)
ez = cpu.zml_batch(bases, offs, threads=4)
print("zml mismatching values:", int((z != ez).sum()), "of", z.size)
p, _ = gpu.query_pml_packed(bases, offs)
ep, _, _ = cpu.pml_batch(bases, offs, threads=4)
print("pml mismatching values:", int((p != ep).sum()), "of", p.size)
