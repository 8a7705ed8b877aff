#!/bin/bash
# round 5: `movi query --no-output` on 1 M x 150 bp -- the scan-ahead helper: windows in flight (MOVI_SCAN_DEPTH) and scanners per window
# (MOVI_SCAN_PARTS), per-chunk parse phases with the wait for the helper and the helper's own time
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_scan${1:+_$1}; mkdir -p $O
IDX=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/short.fa', a)
PY
run() { local name=$1 rep; shift
  for rep in 1 2 3 4 5; do
    movi_amd/bin/movi query -i $IDX --verbose "$@" 2> $O/$name.$rep.err > /dev/null
    grep -h "processing the reads\|Chunk phases" $O/$name.$rep.err | sed "s|^|$name.$rep: |; s/\[movi\] //"
  done
}
{
for round in 1 2; do
run parts1_r$round -r /tmp/short.fa --no-output
MOVI_SCAN_PARTS=2 run parts2_r$round -r /tmp/short.fa --no-output
MOVI_SCAN_PARTS=4 run parts4_r$round -r /tmp/short.fa --no-output
MOVI_SCAN_DEPTH=3 run depth3_r$round -r /tmp/short.fa --no-output
done
} 2>&1 | tee $O/summary.txt
