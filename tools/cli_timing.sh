#!/bin/bash
# End-to-end timing of the host CLI on the cached pangenome workload (run after bench.py built the cache).
D=/tmp/movi_bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np
d="/tmp/movi_bench_cache/pg_5000000_64_0.001_11_m6"
r=np.fromfile(d+"/reads_1000000x150_0.01.bin",np.uint8).reshape(-1,150)
with open("/tmp/reads150.fa","wb") as f:
    for i in range(r.shape[0]):
        f.write(b">r%d\n"%i); f.write(r[i].tobytes()); f.write(b"\n")
PY
ls -la /tmp/reads150.fa
for flags in "--no-output" "-o /tmp/out_a" "-o /tmp/out_b -n"; do
  echo "== movi query $flags"
  ( time ./movi_amd/bin/movi query -i $D -r /tmp/reads150.fa $flags ) 2>&1 | grep -E "Time measured|real|reads are"
done
ls -la /tmp/out_a.pml.bpf
( time ./movi_amd/bin/movi view --bpf /tmp/out_a.pml.bpf > /tmp/view.txt ) 2>&1 | grep real
