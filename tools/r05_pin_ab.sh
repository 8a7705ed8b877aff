#!/bin/bash
# round 5: page-locked read buffers (default) against pageable ones (MOVI_PINNED=0) on the shapes tools/r05_stall.sh does not cover:
# 100 k x 10 kbp reads, and `--gpus 2` on one device
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_pin_ab${1:+_$1}; mkdir -p $O
IDX=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys, os, subprocess
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/short.fa', a)
t = '/tmp/c2_text.bin'
subprocess.check_call(['tools/build_index', 'pangenome', '5000000', '64', '0.001', '11', '6', '/tmp/c2txt', 'text-only'], stderr=subprocess.DEVNULL)
os.rename('/tmp/c2txt/text.bin', t)
subprocess.check_call(['tools/build_index', 'reads', t, '100000', '10000', '0.08', '1011', '/tmp/long.bin'])
bench.write_fasta('/tmp/long.fa', np.fromfile('/tmp/long.bin', np.uint8).reshape(-1, 10000))
PY
run() { local name=$1 rep; shift
  for rep in 1 2 3 4 5; do
    rm -f /tmp/ab_out*
    movi_amd/bin/movi query -i $IDX --verbose "$@" 2> $O/$name.$rep.err > /dev/null
    grep -h "processing the reads\|Stage times" $O/$name.$rep.err | sed "s|^|$name.$rep: |; s/\[movi\] //; s/Time measured for processing the reads: /T /"
  done
}
{
for round in 1 2; do
run long_noout_pinned_r$round -r /tmp/long.fa --no-output
MOVI_PINNED=0 run long_noout_pageable_r$round -r /tmp/long.fa --no-output
done
run long_bpf_pinned -r /tmp/long.fa -o /tmp/ab_out
MOVI_PINNED=0 run long_bpf_pageable -r /tmp/long.fa -o /tmp/ab_out
MOVI_SHARE_GPU=1 run gpus2_pinned -r /tmp/short.fa --no-output --gpus 2
MOVI_SHARE_GPU=1 MOVI_PINNED=0 run gpus2_pageable -r /tmp/short.fa --no-output --gpus 2
} 2>&1 | tee $O/summary.txt
