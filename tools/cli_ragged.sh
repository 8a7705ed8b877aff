#!/bin/bash
# `movi query` end to end on Nanopore-like reads (log-normal lengths around 10 kbp, 8 % substitutions) of a small synthetic
# pangenome: the segment-parallel walk (default) against one lane per read (--seg-len 0); the BPF files must be identical.
set -e
D=/tmp/cli_ragged
mkdir -p $D
[ -f $D/idx/index.movi ] || ./tools/build_index pangenome 500000 64 0.001 11 6 $D/idx 2>/dev/null
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(5)
text = np.fromfile("/tmp/cli_ragged/idx/text.bin", np.uint8)
n = 20000
lens = np.clip(rng.lognormal(np.log(10000) - 0.08, 0.6, n), 200, 200000).astype(np.int64)
with open("/tmp/cli_ragged/reads.fa", "wb") as f:
    for i, L in enumerate(lens):
        s = int(rng.integers(0, text.size - L))
        r = text[s:s + L].copy()
        m = rng.random(L) < 0.08
        r[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(m.sum()))]
        f.write(b">r%d\n" % i); f.write(r.tobytes()); f.write(b"\n")
print("reads: %d, %.1f Mbases, longest %d" % (n, lens.sum() / 1e6, lens.max()))
PY
for flags in "--no-output" "--no-output --seg-len 0" "-o $D/a" "-o $D/b --seg-len 0"; do
  echo "== movi query $flags"
  ( time ./movi_amd/bin/movi query -i $D/idx -r $D/reads.fa $flags ) 2>&1 | grep -E "Time measured for processing|real|rror"
done
cmp $D/a.pml.bpf $D/b.pml.bpf && echo "BPF files identical ($(stat -c %s $D/a.pml.bpf) bytes)"
