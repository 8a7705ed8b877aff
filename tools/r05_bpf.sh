#!/bin/bash
# round 5: the BPF file of `movi query` -- the chunk's records gathered by a pool and written behind it (BpfWriter::append(Chunk)) against
# the one-thread loop (MOVI_BPF_SERIAL=1), 1 M x 150 bp and 100 k x 10 kbp on the c2 index; stage times under --verbose, md5 of the files.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_bpf${1:+_$1}; mkdir -p $O
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
  for rep in 1 2 3; do
    mkdir -p /tmp/bpf_prev; mv -f /tmp/out_*.bpf /tmp/bpf_prev/ 2>/dev/null   # a fresh file every run: re-opening one with O_TRUNC makes ext4 flush it in close() (+30 ms)
    movi_amd/bin/movi query -i $IDX --verbose "$@" 2> $O/$name.$rep.err > /dev/null
    grep -h "processing the reads\|Stage times\|BPF writer" $O/$name.$rep.err | sed "s|^|$name.$rep: |"
  done
}
{
run short_bpf -r /tmp/short.fa -o /tmp/out_short; md5sum /tmp/out_short*.bpf
MOVI_BPF_SERIAL=1 run short_bpf_serial -r /tmp/short.fa -o /tmp/out_short; md5sum /tmp/out_short*.bpf
MOVI_BPF_SLAB_BYTES=2097152 MOVI_BPF_SLABS=8 run short_bpf_slab2x8 -r /tmp/short.fa -o /tmp/out_short
MOVI_BPF_SLAB_BYTES=4194304 run short_bpf_slab4x4 -r /tmp/short.fa -o /tmp/out_short
MOVI_BPF_SLAB_BYTES=4194304 MOVI_BPF_SLABS=8 run short_bpf_slab4x8 -r /tmp/short.fa -o /tmp/out_short
MOVI_BPF_SLAB_BYTES=1048576 MOVI_BPF_SLABS=16 run short_bpf_slab1x16 -r /tmp/short.fa -o /tmp/out_short
MOVI_BPF_SLAB_BYTES=4194304 MOVI_BPF_THREADS=4 run short_bpf_slab4x4_t4 -r /tmp/short.fa -o /tmp/out_short
run long_bpf -r /tmp/long.fa -o /tmp/out_long; md5sum /tmp/out_long*.bpf
MOVI_BPF_SERIAL=1 run long_bpf_serial -r /tmp/long.fa -o /tmp/out_long; md5sum /tmp/out_long*.bpf
MOVI_BPF_SLAB_BYTES=2097152 MOVI_BPF_SLABS=8 run long_bpf_slab2x8 -r /tmp/long.fa -o /tmp/out_long
MOVI_BPF_SLAB_BYTES=4194304 run long_bpf_slab4x4 -r /tmp/long.fa -o /tmp/out_long
MOVI_BPF_SLAB_BYTES=4194304 MOVI_BPF_SLABS=8 run long_bpf_slab4x8 -r /tmp/long.fa -o /tmp/out_long
} 2>&1 | tee $O/summary.txt
timeout 900 python3 -m pytest tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
