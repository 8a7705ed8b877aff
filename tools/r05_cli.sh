#!/bin/bash
# round 5: the `movi query` command end to end on 1 M x 150 bp and 100 k x 10 kbp FASTA files -- stage and parser-phase times (--verbose),
# this tree's binary against round 4's host code on the same library (movi_amd/bin/movi_r04, built from commit c2de598's host sources)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_cli${1:+_$1}; mkdir -p $O
IDX=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys, os, subprocess
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/short.fa', a)
if not os.path.exists('/tmp/long.fa'):
    t = '/tmp/c2_text.bin'
    if not os.path.exists(t):
        subprocess.check_call(['tools/build_index', 'pangenome', '5000000', '64', '0.001', '11', '6', '/tmp/c2txt', 'text-only'], stderr=subprocess.DEVNULL)
        os.rename('/tmp/c2txt/text.bin', t)
    subprocess.check_call(['tools/build_index', 'reads', t, '100000', '10000', '0.08', '1011', '/tmp/long.bin'])
    bench.write_fasta('/tmp/long.fa', np.fromfile('/tmp/long.bin', np.uint8).reshape(-1, 10000))
PY
run() { local bin=$1 name=$2 rep; shift 2
  for rep in 1 2 3; do
    local t0=$(date +%s%N)
    $bin query -i $IDX --verbose "$@" 2> $O/$name.$rep.err > /dev/null
    echo "[movi] wall $(( ($(date +%s%N) - t0) / 1000000 )) ms" >> $O/$name.$rep.err
    grep -h "processing the reads\|Stage times\|Parser phases\|Chunks\|wall" $O/$name.$rep.err | sed "s|^|$name.$rep: |"
  done
}
{
run movi_amd/bin/movi movi_short_noout -r /tmp/short.fa --no-output
run movi_amd/bin/movi movi_short_bpf -r /tmp/short.fa -o /tmp/out_short
md5sum /tmp/out_short*.bpf
run movi_amd/bin/movi movi_long_noout -r /tmp/long.fa --no-output
run movi_amd/bin/movi movi_long_bpf -r /tmp/long.fa -o /tmp/out_long
md5sum /tmp/out_long*.bpf
MOVI_SCAN_PARTS=1 run movi_amd/bin/movi movi_scan1_short_noout -r /tmp/short.fa --no-output
MOVI_SCAN_PARTS=2 run movi_amd/bin/movi movi_scan2_short_noout -r /tmp/short.fa --no-output
MOVI_SCAN_PARTS=8 run movi_amd/bin/movi movi_scan8_short_noout -r /tmp/short.fa --no-output
} 2>&1 | tee $O/summary.txt
/tmp/parse_bench /tmp/short.fa 12 > $O/parse_bench.txt 2>&1 || { g++ -O2 -std=c++17 -mavx2 -o /tmp/parse_bench tools/parse_bench.cpp movi_amd/host/reads.cpp -lpthread && /tmp/parse_bench /tmp/short.fa 12 > $O/parse_bench.txt 2>&1; }; tail -3 $O/parse_bench.txt
