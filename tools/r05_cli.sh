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
run() { exe=$1; name=$2; shift 2
  for rep in 1 2 3; do
    /usr/bin/time -f "wall %e s" $exe query -i $IDX --verbose "$@" 2> $O/$name.$rep.err > /dev/null
    grep -h "processing the reads\|Stage times\|Parser phases\|wall" $O/$name.$rep.err | sed "s/^/$name.$rep: /"
  done
}
{
for exe in movi movi_r04; do
  run movi_amd/bin/$exe ${exe}_short_noout -r /tmp/short.fa --no-output
  run movi_amd/bin/$exe ${exe}_short_bpf -r /tmp/short.fa -o /tmp/out_short
  run movi_amd/bin/$exe ${exe}_long_noout -r /tmp/long.fa --no-output
  run movi_amd/bin/$exe ${exe}_long_bpf -r /tmp/long.fa -o /tmp/out_long
  md5sum /tmp/out_short*.bpf /tmp/out_long*.bpf | sed "s/^/$exe: /"
done
MOVI_NO_CHUNK_RAMP=1 run movi_amd/bin/movi movi_noramp_short_noout -r /tmp/short.fa --no-output
MOVI_NO_AFFINITY=1 run movi_amd/bin/movi movi_noaff_short_noout -r /tmp/short.fa --no-output
for t in 6 8 12; do MOVI_PARSE_THREADS=$t run movi_amd/bin/movi movi_t${t}_short_noout -r /tmp/short.fa --no-output; done
MOVI_PARSE_THREADS=8 MOVI_NO_AFFINITY=1 run movi_amd/bin/movi movi_t8_noaff_short_noout -r /tmp/short.fa --no-output
} 2>&1 | tee $O/summary.txt
