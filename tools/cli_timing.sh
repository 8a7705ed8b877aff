#!/bin/bash
# End-to-end timing of the host CLI on the cached pangenome workload (run after bench.py built the cache:
# `python bench.py --steps 2 --no-cpu-baseline` and `--workload c3`).
D=/tmp/movi_bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import glob, numpy as np
d="/tmp/movi_bench_cache/pg_5000000_64_0.001_11_m6"
for pat, L, out in (("reads_*x150_*.bin", 150, "/tmp/reads150.fa"), ("reads_*x10000_*.bin", 10000, "/tmp/reads10k.fa")):
    fs = sorted(glob.glob(d + "/" + pat))
    if not fs:
        continue
    r = np.fromfile(fs[0], np.uint8).reshape(-1, L)
    with open(out, "wb") as f:
        for i in range(r.shape[0]):
            f.write(b">r%d\n" % i); f.write(r[i].tobytes()); f.write(b"\n")
PY
for R in /tmp/reads150.fa /tmp/reads10k.fa; do
  [ -f $R ] || continue
  ls -la $R
  for flags in "--no-output" "-o /tmp/out_a" "-o /tmp/out_b -n" "--count -o /tmp/out_c"; do
    echo "== movi query -r $R $flags"
    ( time ./movi_amd/bin/movi query -i $D -r $R --verbose $flags ) 2>&1 | grep -E "Time measured for processing|Stage times|Parser phases|real|Error"
  done
done
ls -la /tmp/out_a.pml.bpf
( time ./movi_amd/bin/movi view --bpf /tmp/out_a.pml.bpf > /tmp/view.txt ) 2>&1 | grep real
