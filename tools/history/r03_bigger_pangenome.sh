#!/bin/bash
# a 2.5 x bigger pangenome than c2 (12.5 Mbp ancestor x 64 genomes x 2 strands = 1.6 Gbp, ~35 M rows: 280 MB of rows, beyond the
# Infinity Cache): plain rows / look-ahead rows / fat rows on a REAL BWT of that size
set -e
D=/tmp/pg_big
mkdir -p $D gpurun_out/r04r
g++ -O2 -std=c++17 -o /tmp/build_index tools/build_index.cpp
( time /tmp/build_index pangenome 12500000 64 0.001 11 6 $D ) 2>&1 | tail -4
/tmp/build_index reads $D/text.bin 1000000 150 0.01 11 $D/reads.bin
ls -la $D
for ah in 0 1 2; do
  python3 bench.py --from-dir $D --quick --steps 10 --warmup 3 --ahead-rows $ah 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; l=d['roofline']['launch']
print('ahead_rows $ah ->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3), 'ms', 'rows', c['rows'], 'it/base', c.get('iterations_per_base'), 'cap', l['waves_per_cu'], 'ahead', l['ahead'])"
done | tee gpurun_out/r04r/big_pangenome.txt
python3 tools/lf_chain_stats.py $D/index.movi | tee -a gpurun_out/r04r/big_pangenome.txt
