#!/bin/bash
# round 5: the host I/O primitives under `movi query` on the GPU box's CPU (tools/io_bench.cpp): file mapping against pread for the read
# file, write() stream against parallel pwrite / shared mapping / O_DIRECT for the BPF file.  No GPU work.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_io${1:+_$1}; mkdir -p $O
g++ -O2 -std=c++17 -pthread -o /tmp/io_bench tools/io_bench.cpp || exit 1
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/short.fa', a)
PY
{
nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/kernel/mm/transparent_hugepage/enabled; df -T /tmp | tail -1; uname -r
/tmp/io_bench read /tmp/short.fa
/tmp/io_bench write /tmp/io_bench_out.bin 320
echo "# /dev/shm"; /tmp/io_bench write /dev/shm/io_bench_out.bin 320
} 2>&1 | tee $O/summary.txt
